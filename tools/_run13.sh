export TMPDIR=/tmp
O=gpurun_out/r04m; mkdir -p $O
B="--batch 9216 --steps 30 --warmup 3 --cpu-seconds 0 --lt= --variant-keys 0 --stream-keys 0 --key-per-item 0 --lt-direct 0"
one() { printf "%-58s " "$*" >> $O/chunks.txt; env "$@" timeout 300 python bench.py $B $SET 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.readline()); k=d['roofline']['kernel_avg_us']; print(round(d['value']), d['verified'], {n[3:-7]: round(v,1) for n,v in k.items() if v})" >> $O/chunks.txt; }
SET="--set C2"; for c in 512 768 1024; do one HEFX_CHUNK=$c; done
one HEFX_CHUNK=1024 HEFX_STREAMS=3
one HEFX_CHUNK=768 HEFX_STREAM_X=0
one HEFX_CHUNK=512
cat $O/chunks.txt
python tools/make_lr_csv.py 2000 drivers/_ref/pulsar_stars_copy.csv
cd drivers/_ref; ulimit -c 0
for i in $(seq 1 16); do
  s=$(date +%s.%N); timeout 120 ./logistic_regression_ckks > /dev/null 2> ../../$O/err$i.txt; e=$(date +%s.%N)
  echo "run $i: $(echo "$e - $s" | bc) s  $(grep hefx ../../$O/err$i.txt | head -3)"
done
rm -f pulsar_stars_copy.csv
