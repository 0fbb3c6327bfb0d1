#!/bin/bash
# Same-box A/B of library builds over the bench sets (development aid, GPU box):
#   tools/ab_sweep.sh A default      (build/libhefx_A.so against the in-tree library), two rounds each, interleaved
for round in 1 2; do
for set in C3 C2 C4 C5; do
for tag in "$@"; do
  if [ "$tag" = default ]; then lib=""; else lib="$PWD/build/libhefx_$tag.so"; fi
  printf "%s %-4s %-10s " $round $set "$tag"
  case $set in C3) b=4608;; C2) b=9216;; C4) b=2304;; C5) b=2304;; esac
  HEFX_LIB=$lib python bench.py --set $set --batch $b --steps 30 --warmup 3 --quick 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.readline()); k=d['roofline']['kernel_avg_us']; print(round(d['value']), d['verified'], {n[3:-7]: round(v) for n,v in k.items() if v})"
done; done; done
