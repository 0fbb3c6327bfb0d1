export TMPDIR=/tmp
O=gpurun_out/r04d; mkdir -p $O
python -m pytest tests -x -q -m gpu > $O/gpu_tests.txt 2>&1; tail -5 $O/gpu_tests.txt
tools/lr_driver_profile.sh $O/lr_before > $O/lr_before.log 2>&1
cat $O/lr_before/run*.txt | grep -E "real|recorded"
