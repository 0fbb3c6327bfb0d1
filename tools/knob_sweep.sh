#!/bin/bash
# Sweeps the launch-structure knobs of the key switch on the bench workload (development aid; run on the GPU box):
#   tools/knob_sweep.sh > gpurun_out/knobs.txt
run() { printf "%-40s " "$*"; env "$@" python bench.py --steps 30 --warmup 2 --cpu-seconds 0 --lt "" --variant-keys 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(round(d['value']))"; }
run HEFX_NOP=1
run HEFX_STREAMS=3
run HEFX_STREAMS=4
run HEFX_STREAMS=0
run HEFX_CHUNK=128
run HEFX_CHUNK=192
run HEFX_CHUNK=224
run HEFX_STREAM_X=0
run HEFX_CHUNK=128 HEFX_STREAMS=3
run HEFX_CHUNK=128 HEFX_STREAM_X=0
run HEFX_CHUNK=64 HEFX_STREAM_X=0 HEFX_STREAMS=4
