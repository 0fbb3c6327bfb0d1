export TMPDIR=/tmp
O=gpurun_out/r04final; mkdir -p $O
timeout 1500 python -m pytest tests -x -q -m gpu > $O/gpu_tests.txt 2>&1; tail -4 $O/gpu_tests.txt
timeout 1800 tools/round_profiles.sh $O > $O/round_profiles.log 2>&1
tools/lr_driver_profile.sh $O/lr_driver 8 > $O/lr_driver.log 2>&1; cat $O/lr_driver.log
head -c 600 $O/bench.json
