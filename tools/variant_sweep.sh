#!/bin/bash
# Runs the bench workload with alternative builds of the library (tools/build_variant.sh; development aid, GPU box):
#   tools/variant_sweep.sh default epi8 pipe4 ...
for tag in "$@"; do
  if [ "$tag" = default ]; then lib=""; else lib="$PWD/build/libhefx_$tag.so"; fi
  printf "%-14s " "$tag"
  HEFX_LIB=$lib python bench.py --steps 30 --warmup 2 --cpu-seconds 0 --lt= --variant-keys 0 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.readline()); k=d['roofline']['kernel_avg_us']; print(round(d['value']), d['verified'], 'finish_us', round(k['ks_moddown_finish_kernel'],1), 'ntt_us', round(k['ks_ntt_digits_kernel'],1))"
done
