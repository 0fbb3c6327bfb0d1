#!/bin/bash
# The reference's logistic_regression_ckks.cpp, unchanged (drivers/_ref), on 2000 synthetic pulsar-shaped rows (GPU box):
#   tools/lr_driver_profile.sh outdir [api]  -> wall time (3 runs), the shim's submission timeline; with `api` also the HIP
#   API statistics (rocprofv3 --hip-runtime-trace --stats: slow, minutes)
out=${1:-gpurun_out/lr_driver}; mkdir -p $out; root=$PWD
export TMPDIR=/tmp
python tools/make_lr_csv.py 2000 drivers/_ref/pulsar_stars_copy.csv
cd drivers/_ref
ulimit -c 0   # the driver ends in std::terminate (SEAL's "scale out of bounds", :336): no core file in the timing
for i in 1 2 3; do ( time SEAL_SHIM_STATS=1 timeout 120 ./logistic_regression_ckks ) 2>&1 | tail -n 12 > $root/$out/run$i.txt; done
( time SEAL_SHIM_STATS=2 timeout 120 ./logistic_regression_ckks ) 2>&1 | grep -v "^|\|^/\|^\\\\" | tail -n 60 > $root/$out/timeline.txt
if [ "$2" = api ]; then
  timeout 600 rocprofv3 --hip-runtime-trace --stats --output-format csv -d $root/$out/hip -o h -- ./logistic_regression_ckks > /dev/null 2>&1
  rm -f $root/$out/hip/*_trace.csv $root/$out/hip/*_agent_info.csv
fi
rm -f pulsar_stars_copy.csv
cd $root
grep -h "real\|recorded" $out/run*.txt
