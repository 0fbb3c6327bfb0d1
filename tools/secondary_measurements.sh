#!/bin/bash
# The composite-level measurements quoted in profiles/README.md, in one go (GPU box): tools/secondary_measurements.sh outdir
out=${1:-gpurun_out/secondary}; mkdir -p $out $out/drivers
python tools/single_op_latency.py C3 > $out/single_op_C3.txt 2>&1
python tools/single_op_latency.py C2 > $out/single_op_C2.txt 2>&1
python tools/small_batch_latency.py > $out/small_batch.txt 2>&1
python tools/lt_bench.py 10 100 1000 > $out/lt_n8192.txt 2>&1
LT_SET=C3 python tools/lt_bench.py 16 128 512 > $out/lt_C3.txt 2>&1
python tools/matmul_bench.py C3 4 dense > $out/mm_C3_4_dense.txt 2>&1
python tools/matmul_bench.py C3 4 sparse > $out/mm_C3_4_sparse.txt 2>&1
python tools/matmul_bench.py C5 64 sparse > $out/mm_C5_64_sparse.txt 2>&1
python tools/matmul_bench.py C5 64 sparse_hoisted > $out/mm_C5_64_sparse_hoisted.txt 2>&1
python tools/lr_bench.py 100 500 2000 > $out/lr.txt 2>&1
LR_LOG_SUM=1 python tools/lr_bench.py 100 500 2000 > $out/lr_fast.txt 2>&1
python tools/encode_bench.py > $out/encode.txt 2>&1
tools/run_reference_drivers.sh $out/drivers > $out/drivers.log 2>&1
# the reference's LR driver over 2000 rows, recorded and call by call
python tools/make_lr_csv.py 2000 drivers/_ref/pulsar_stars_copy.csv
( cd drivers/_ref && ( time SEAL_SHIM_STATS=1 timeout 300 ./logistic_regression_ckks ) > ../../$out/lr_driver_2000.txt 2>&1; ( time SEAL_SHIM_LAZY=0 timeout 300 ./logistic_regression_ckks ) > ../../$out/lr_driver_2000_eager.txt 2>&1; ( time SEAL_SHIM_STATS=1 SEAL_SHIM_DEVICES=2 timeout 300 ./logistic_regression_ckks ) > ../../$out/lr_driver_2000_devices2.txt 2>&1; rm -f pulsar_stars_copy.csv )
for f in $out/lr_driver_2000*.txt; do tail -n 40 $f > $f.tail; mv $f.tail $f; done
ls $out
