#!/bin/bash
# Same-box A/B of ENVIRONMENT settings over bench sets (development aid, GPU box):
#   tools/ab_env.sh "C3 C2" "HEFX_FUSED=0" "HEFX_FUSED=1" ...   two interleaved rounds, one line each
sets=$1; shift
for round in 1 2; do
for set in $sets; do
for e in "$@"; do
  printf "%s %-4s %-28s " $round $set "$e"
  case $set in C3) b=4608;; C2) b=9216;; C4) b=2304;; C5) b=2304;; esac
  env $e python bench.py --set $set --batch $b --steps 20 --warmup 2 --cpu-seconds 0 --lt= --variant-keys 0 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.readline()); k=d['roofline']['kernel_avg_us']; print(round(d['value']), d['verified'], {n[3:-7]: round(v) for n,v in k.items() if v})"
done; done; done
