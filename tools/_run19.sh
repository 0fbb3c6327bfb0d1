cd $GRAFT_REPO_ROOT
( timeout 400 python tools/stress_parity.py 240 2>&1 | tail -3 ) > gpurun_out/stress_default.txt
for e in "HEFX_NO_FP64=1" "HEFX_FUSED=1" "HEFX_QUARTER=1" "HEFX_STREAMS=0 HEFX_CHUNK=16"; do
  ( env $e timeout 200 python tools/stress_parity.py 75 2>&1 | tail -2 | sed "s/^/$e: /" ) >> gpurun_out/stress_knobs.txt
done
for e in "HEFX_FUSED=1" "HEFX_NO_FP64=1" "HEFX_QUARTER=1" "HEFX_STREAMS=0" "HEFX_MAC_X=1"; do
  echo "== suite under $e: $(env $e timeout 900 python -m pytest tests -x -q -m gpu 2>&1 | grep 'passed\|failed' | tail -1)" >> gpurun_out/suite_knobs.txt
done
cat gpurun_out/stress_default.txt gpurun_out/stress_knobs.txt gpurun_out/suite_knobs.txt
