export TMPDIR=/tmp
O=gpurun_out/r04l; mkdir -p $O
which strace perf ltrace 2>&1 | head -3
python tools/make_lr_csv.py 2000 drivers/_ref/pulsar_stars_copy.csv
cd drivers/_ref; ulimit -c 0
for i in $(seq 1 16); do
  /usr/bin/time -f "%e s wall" timeout 120 ./logistic_regression_ckks > /dev/null 2> ../../$O/err$i.txt
  echo "run $i: $(grep 'wall' ../../$O/err$i.txt) $(grep hefx ../../$O/err$i.txt | head -3)"
done
rm -f pulsar_stars_copy.csv
cd ../..
B="--batch 9216 --steps 30 --warmup 3 --cpu-seconds 0 --lt= --variant-keys 0 --stream-keys 0 --key-per-item 0 --lt-direct 0"
one() { printf "%-58s " "$*" >> $O/chunks.txt; env "$@" timeout 300 python bench.py $B $SET 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.readline()); k=d['roofline']['kernel_avg_us']; print(round(d['value']), d['verified'], {n[3:-7]: round(v,1) for n,v in k.items() if v})" >> $O/chunks.txt; }
SET="--set C3"; for c in 256 384 512; do one HEFX_CHUNK=$c; done
SET="--set C4"; for c in 256 384 512; do one HEFX_CHUNK=$c; done
SET="--set C5"; for c in 128 256 384; do one HEFX_CHUNK=$c; done
SET="--set C2"; for c in 256 512; do one HEFX_CHUNK=$c; done
cat $O/chunks.txt
