#!/bin/bash
# End-of-round measurement set (GPU box, from the repo root):  tools/round_profiles.sh gpurun_out/r04final
# Leaves: bench.json (+ C2/C4/C5), kernel-trace stats with and without the two-stream overlap (C3 and C2), the two HBM
# traffic passes folded by tools/pmc_traffic.py (C3 and C2), one SQ counter pass each.  Copy what is to be judged into
# profiles/.
out=${1:-gpurun_out/round}; mkdir -p $out
export TMPDIR=/tmp
python bench.py > $out/bench.json 2> $out/bench.err
python bench.py --set C2 --batch 9216 --cpu-seconds 0 --lt= --lt-direct 0 --key-per-item 0 --secondary= --composites= --ladder= > $out/bench_C2.json 2>/dev/null
python bench.py --set C4 --batch 2304 --cpu-seconds 0 --lt= --lt-direct 0 --key-per-item 0 --secondary= --composites= --ladder= > $out/bench_C4.json 2>/dev/null
python bench.py --set C5 --batch 2304 --cpu-seconds 0 --lt= --lt-direct 0 --key-per-item 0 --secondary= --composites= --ladder= > $out/bench_C5.json 2>/dev/null
for set in C3 C2; do
  case $set in C3) b=4608; per=256;; C2) b=9216; per=512;; esac
  B="--set $set --batch $b --steps 5 --warmup 1 --cpu-seconds 0 --variant-keys 0 --stream-keys 0 --key-per-item 0 --lt-direct 0 --secondary= --sustain 0 --lt= --composites= --ladder="
  P="--set $set --batch $b --steps 2 --warmup 0 --cpu-seconds 0 --variant-keys 0 --stream-keys 0 --key-per-item 0 --lt-direct 0 --secondary= --sustain 0 --lt= --composites= --ladder="
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt_$set -o kt -- python3 bench.py $B > /dev/null 2> $out/kt_$set.err
  HEFX_STREAMS=0 rocprofv3 --kernel-trace --stats --output-format csv -d $out/kts_$set -o kts -- python3 bench.py $B > /dev/null 2> $out/kts_$set.err
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pf_$set -o pf -- python3 bench.py $P > /dev/null 2> $out/pf_$set.err
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/pw_$set -o pw -- python3 bench.py $P > /dev/null 2> $out/pw_$set.err
  python tools/pmc_traffic.py $out/pf_$set $out/pw_$set per-chunk:$per $out/pmc_traffic_$set.json "python3 bench.py $P (2 timed steps, 2 warm steps and 1 profiled pass of $b ops in chunks of $per)" > /dev/null
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_LDS SQ_ACTIVE_INST_ANY --output-format csv -d $out/sq_$set -o sq -- python3 bench.py $P > /dev/null 2> $out/sq_$set.err
  python - $out $set $per <<'PY'
import csv, collections, json, sys, glob, os
out, st, per = sys.argv[1], sys.argv[2], int(sys.argv[3])
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for f in glob.glob(os.path.join(out, "sq_" + st, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("hefx::", "")
        if not k.startswith("ks_"): continue
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == "SQ_INSTS_VALU": n[k] += 1
sys.path.insert(0, os.getcwd())
from seal_fyp_logistic_regression_amd import _build
res = {k: dict(launches=n[k], items_per_launch=per, **v) for k, v in agg.items()}
res["_meta"] = {"csrc_sha16": _build.source_sha16(), "libhefx_sha16": _build.library_sha16()}  # bench.py: stale-counter check
json.dump(res, open(os.path.join(out, f"sq_counters_{st}.json"), "w"), indent=1)
PY
  rm -rf $out/pf_$set $out/pw_$set $out/sq_$set
done
# the exactly hoisted linear transform (d = 512, a direct key per step): kernel stats and the timeline of the last call
rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt_lt_direct -o kt -- python3 tools/lt_direct_probe.py 512 10 > $out/lt_direct_probe.txt 2> $out/kt_lt_direct.err
python tools/kernel_timeline.py $out/kt_lt_direct > $out/lt_direct_timeline.txt
find $out -name "*_agent_info.csv" -delete; find $out -name "*kernel_trace.csv" -delete
python tools/batch_sweep.py C3 > $out/batch_sweep_C3.json 2> $out/batch_sweep.err
ls $out
