export TMPDIR=/tmp
O=gpurun_out/r04j; mkdir -p $O
timeout 120 drivers/_ref/shim_selftest > $O/selftest.txt 2>&1; grep -v "^ok" $O/selftest.txt | tail -5
tools/lr_driver_profile.sh $O/lr > $O/lr.log 2>&1; tail -4 $O/lr.log; grep "seal shim" $O/lr/timeline.txt | cut -c1-150
for i in 1 2 3; do tools/lr_driver_profile.sh $O/lr_again$i > $O/lr_again$i.log 2>&1; tail -3 $O/lr_again$i.log; done
B="--set C2 --batch 9216 --steps 30 --warmup 3 --cpu-seconds 0 --lt= --variant-keys 0 --stream-keys 0 --key-per-item 0 --lt-direct 0"
one() { printf "%-58s " "$*" >> $O/c2.txt; env "$@" timeout 300 python bench.py $B 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.readline()); k=d['roofline']['kernel_avg_us']; print(round(d['value']), d['verified'], {n[3:-7]: round(v,1) for n,v in k.items() if v})" >> $O/c2.txt; }
for round in 1 2; do
one X=0
one HEFX_LIB=build/libhefx_w2_13.so
one HEFX_LIB=build/libhefx_w3_13.so
one HEFX_LIB=build/libhefx_w4_13.so
done
one HEFX_CHUNK=384
one HEFX_CHUNK=512
one HEFX_CHUNK=512 HEFX_STREAM_X=0
one HEFX_CHUNK=512 HEFX_SUB=256
one HEFX_STREAMS=3
one HEFX_CHUNK=192
cat $O/c2.txt
