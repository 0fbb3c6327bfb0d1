export TMPDIR=/tmp
O=gpurun_out/r04g; mkdir -p $O
tools/lr_driver_profile.sh $O/lr_default > $O/lr_default.log 2>&1; echo default; tail -3 $O/lr_default.log; grep "seal shim" $O/lr_default/timeline.txt | cut -c1-150
for mb in 2048 4096 16384; do SEAL_SHIM_PENDING_MB=$mb tools/lr_driver_profile.sh $O/lr_mb$mb > $O/lr_mb$mb.log 2>&1; echo "pending_mb=$mb"; tail -3 $O/lr_mb$mb.log; done
timeout 1200 python -m pytest tests -x -q -m gpu > $O/gpu_tests.txt 2>&1; tail -6 $O/gpu_tests.txt
