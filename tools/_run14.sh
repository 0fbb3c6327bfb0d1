export TMPDIR=/tmp
O=gpurun_out/r04n; mkdir -p $O
for e in "X=0" "HEFX_CHUNK=128" "HEFX_CHUNK=64" "HEFX_CHUNK=128 HEFX_STREAMS=3" "HEFX_CHUNK=128 HEFX_STREAMS=4" "HEFX_CHUNK=192 HEFX_STREAMS=3"; do
  echo "== $e" >> $O/kpi.txt
  env $e python tools/key_per_item_probe.py 1024 5 2>&1 | grep "key-per-item" >> $O/kpi.txt
  env $e python tools/lt_direct_probe.py 512 10 >> $O/kpi.txt 2>&1
done
cat $O/kpi.txt
