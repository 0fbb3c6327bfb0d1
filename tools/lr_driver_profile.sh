#!/bin/bash
# The reference's logistic_regression_ckks.cpp, unchanged (drivers/_ref), on 2000 synthetic pulsar-shaped rows (GPU box):
#   tools/lr_driver_profile.sh outdir [runs=5]  -> wall time of every run and the shim's submission timeline of each
# (no rocprofv3 here: the driver ends in std::terminate, and the profiler does not come back from that)
out=${1:-gpurun_out/lr_driver}; runs=${2:-5}; mkdir -p $out; root=$PWD
python tools/make_lr_csv.py 2000 drivers/_ref/pulsar_stars_copy.csv
cd drivers/_ref
ulimit -c 0   # the driver ends in std::terminate (SEAL's "scale out of bounds", :336): no core file in the timing
for i in $(seq 1 $runs); do
  ( time SEAL_SHIM_STATS=2 timeout 120 ./logistic_regression_ckks ) > $root/$out/run$i.full 2>&1
  grep "seal shim\|^real\|^user\|^sys" $root/$out/run$i.full | cut -c1-170 > $root/$out/run$i.txt; rm -f $root/$out/run$i.full
done
rm -f pulsar_stars_copy.csv
cd $root
grep -h "^real" $out/run*.txt | tr '\n' ' '; echo
