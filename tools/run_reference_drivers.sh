#!/bin/bash
# Runs the reference's own drivers (built unchanged by `make -C drivers`) on the GPU box and captures their output
# the way profiles/r01_reference_drivers/ holds it (long outputs: first and last 60 lines).
# usage (on the GPU box, from the repo root): tools/run_reference_drivers.sh gpurun_out/drivers
out=${1:-gpurun_out/drivers}; mkdir -p $out; root=$PWD
clip() { awk -v n=60 '{a[NR]=$0} END{ if (NR<=2*n) {for(i=1;i<=NR;i++) print a[i]} else {for(i=1;i<=n;i++) print a[i]; print "[...]"; for(i=NR-n+1;i<=NR;i++) print a[i]} }'; }
cd drivers/_ref
export SEAL_SHIM_SYNC=1   # the drivers time their calls with chrono: make every shim call wait for the device
for d in 4_ckks linear_transformation2 matrix_transpose linear_transformation matrix_mult_benchmark 5_rotation vector_ops \
         1_bfv 2_encoders 3_levels benchmark benchmark2 matrix_ops; do
  timeout 120 ./$d 2>&1 | clip > $root/$out/$d.txt
done
( time timeout 120 ./matrix_multiplication ) 2>&1 | clip > $root/$out/matrix_multiplication.txt
# degree 3, x = 0.5, Horner, Tree, quit
printf '3\n0.5\n1\n2\n0\n' | timeout 120 ./polynomial 2>&1 | head -c 200000 | clip > $root/$out/polynomial.txt
cp $root/tests/golden/pulsar_rows_head400.csv pulsar_stars_copy.csv
timeout 300 ./logistic_regression_ckks 2>&1 | clip > $root/$out/logistic_regression_ckks.txt
rm -f pulsar_stars_copy.csv *.dat script_*.p *.p 2>/dev/null
