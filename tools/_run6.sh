export TMPDIR=/tmp
O=gpurun_out/r04f; mkdir -p $O
for g in 1 0; do
  HEFX_CHAIN_GRAPH=$g timeout 300 python tools/chain_latency.py 8 2 500 >> $O/chain.txt 2>&1
  HEFX_CHAIN_GRAPH=$g timeout 300 python tools/chain_latency.py 8 7 200 >> $O/chain.txt 2>&1
  HEFX_CHAIN_GRAPH=$g timeout 300 python tools/chain_latency.py 1 2 500 >> $O/chain.txt 2>&1
done
cat $O/chain.txt
timeout 120 drivers/_ref/shim_selftest > $O/selftest.txt 2>&1; grep -v "^ok" $O/selftest.txt | tail -5
for g in 1 0; do HEFX_CHAIN_GRAPH=$g tools/lr_driver_profile.sh $O/lr_graph$g > $O/lr_graph$g.log 2>&1; echo "graph=$g"; cat $O/lr_graph$g.log | tail -4; grep "seal shim" $O/lr_graph$g/timeline.txt | cut -c1-130 | tail -4; done
