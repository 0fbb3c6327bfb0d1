export TMPDIR=/tmp
O=gpurun_out/r04h; mkdir -p $O
B="--set C2 --batch 9216 --steps 30 --warmup 3 --cpu-seconds 0 --lt= --variant-keys 0 --stream-keys 0 --key-per-item 0 --lt-direct 0"
one() { printf "%-44s " "$*" >> $O/c2.txt; env "$@" python bench.py $B 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.readline()); k=d['roofline']['kernel_avg_us']; print(round(d['value']), d['verified'], {n[3:-7]: round(v,1) for n,v in k.items() if v})" >> $O/c2.txt; }
for round in 1 2; do
one X=0
one HEFX_LIB=build/libhefx_w2_13.so
one HEFX_LIB=build/libhefx_w3_13.so
one HEFX_LIB=build/libhefx_w4_13.so
done
one HEFX_STREAMS=3
one HEFX_STREAMS=4
one HEFX_CHUNK=192
one HEFX_CHUNK=128
one HEFX_STREAM_X=1
cat $O/c2.txt
