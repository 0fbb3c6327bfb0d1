#!/bin/bash
# Engine clock and board power while the bench workload runs (GPU box): samples rocm-smi once a second next to a
# long bench run.  usage: tools/clock_under_load.sh [outfile]
out=${1:-gpurun_out/clock_under_load.txt}
python bench.py --steps 900 --warmup 3 --cpu-seconds 0 --lt= --variant-keys 0 --stream-keys 0 > /tmp/bench_long.json 2>/dev/null &
pid=$!
sleep 5   # context creation + input sampling
{
echo "# rocm-smi samples during: python bench.py --steps 900 (C3, 4608 ops per step, ~17 s of load)"
for i in 1 2 3 4 5 6 7 8 9 10; do
  /opt/rocm/bin/rocm-smi --showclocks --showpower --showuse 2>/dev/null | grep -iE "sclk|mclk|power|busy|use" | tr -s ' ' | sed "s/^/t=$i /"
  sleep 1
done
} > $out
wait $pid
python -c "import json; d=json.load(open('/tmp/bench_long.json')); print('# bench value', round(d['value']), 'ops/s, ms_per_step', round(d['ms_per_step'],3))" >> $out
cat $out
