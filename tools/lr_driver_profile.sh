#!/bin/bash
# The reference's logistic_regression_ckks.cpp, unchanged (drivers/_ref), on 2000 synthetic pulsar-shaped rows (GPU box):
#   tools/lr_driver_profile.sh outdir [runs=5] [pause_s=0]  -> wall time of every run and the shim's submission timeline of each
# pause_s: seconds to wait between runs.  Back to back, a run's first large hipMalloc waits for the driver to scrub the ~16 GB
# the PREVIOUS process has just freed (0.8-0.9 s with the process asleep: profiles/r05/lr_driver_2000/README.txt); two
# seconds of pause take that out of the measurement.
# (no rocprofv3 here: the driver ends in std::terminate, and the profiler does not come back from that)
out=${1:-gpurun_out/lr_driver}; runs=${2:-5}; pause=${3:-0}; mkdir -p $out; root=$PWD
python tools/make_lr_csv.py 2000 drivers/_ref/pulsar_stars_copy.csv
cd drivers/_ref
ulimit -c 0   # the driver ends in std::terminate (SEAL's "scale out of bounds", :336): no core file in the timing
for i in $(seq 1 $runs); do
  sleep $pause
  ( time SEAL_SHIM_STATS=2 timeout 120 ./logistic_regression_ckks ) > $root/$out/run$i.full 2>&1
  grep "seal shim\|hefx\] hipMalloc\|^real\|^user\|^sys" $root/$out/run$i.full | cut -c1-170 > $root/$out/run$i.txt; rm -f $root/$out/run$i.full
done
rm -f pulsar_stars_copy.csv
cd $root
grep -h "^real" $out/run*.txt | tr '\n' ' '; echo
