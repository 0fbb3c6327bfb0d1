#!/bin/bash
# Occupancy-cap experiment (development aid, GPU box): the bench sets with a larger LDS request per NTT kernel
# (HEFX_{FIN,NTT,INTT,MDI}_LDS bytes -> fewer workgroups per CU).  usage: tools/occupancy_sweep.sh C3 4608
set=${1:-C3}; b=${2:-4608}
run() { printf "%-44s " "$*"; env "$@" python bench.py --set $set --batch $b --steps 20 --warmup 2 --cpu-seconds 0 --lt= --variant-keys 0 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.readline()); k=d['roofline']['kernel_avg_us']; print(round(d['value']), {n[3:-7]: round(v) for n,v in k.items() if v})"; }
run X=0
for v in 60000 90000; do for k in FIN NTT INTT MDI; do run HEFX_${k}_LDS=$v; done; done
