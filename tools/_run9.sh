export TMPDIR=/tmp
O=gpurun_out/r04i; mkdir -p $O
tools/lr_driver_profile.sh $O/lr_16g api > $O/lr_16g.log 2>&1; tail -3 $O/lr_16g.log; grep "seal shim" $O/lr_16g/timeline.txt | cut -c1-150
python - <<'PY'
import csv
rows=list(csv.DictReader(open('gpurun_out/r04i/lr_16g/hip/h_hip_api_stats.csv')))
rows.sort(key=lambda r:-float(r['TotalDurationNs']))
print("total HIP API time %.3f s"%(sum(float(r['TotalDurationNs']) for r in rows)/1e9))
for r in rows[:14]:
    print(f"{r['Name']:40s} calls {r['Calls']:>7s} total_ms {float(r['TotalDurationNs'])/1e6:9.1f} avg_us {float(r['AverageNs'])/1e3:8.1f}")
PY
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
