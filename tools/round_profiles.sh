#!/bin/bash
# End-of-round measurement set (GPU box, from the repo root):  tools/round_profiles.sh gpurun_out/r02final
# Leaves: bench.json (+ C2/C4/C5), kernel-trace stats with and without the two-stream overlap, the two HBM traffic
# passes folded by tools/pmc_traffic.py, one SQ counter pass.  Copy what is to be judged into profiles/.
out=${1:-gpurun_out/round}; mkdir -p $out
export TMPDIR=/tmp
B="--steps 5 --warmup 1 --cpu-seconds 0 --variant-keys 0 --stream-keys 0 --lt="
python bench.py > $out/bench.json 2> $out/bench.err
python bench.py --set C2 --batch 9216 --cpu-seconds 0 --lt= > $out/bench_C2.json 2>/dev/null
python bench.py --set C4 --batch 2304 --cpu-seconds 0 --lt= > $out/bench_C4.json 2>/dev/null
python bench.py --set C5 --batch 2304 --cpu-seconds 0 --lt= > $out/bench_C5.json 2>/dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt -o kt -- python3 bench.py $B > $out/kt.out 2> $out/kt.err
HEFX_STREAMS=0 rocprofv3 --kernel-trace --stats --output-format csv -d $out/kts -o kts -- python3 bench.py $B > $out/kts.out 2> $out/kts.err
P="--steps 2 --warmup 0 --cpu-seconds 0 --variant-keys 0 --stream-keys 0 --lt="
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pf -o pf -- python3 bench.py $P > /dev/null 2> $out/pf.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/pw -o pw -- python3 bench.py $P > /dev/null 2> $out/pw.err
python tools/pmc_traffic.py $out/pf $out/pw per-chunk:256 $out/pmc_traffic.json "python3 bench.py $P (2 timed steps, 2 warm steps and 1 profiled pass of 4608 ops = 18 chunks of 256 each)" > /dev/null
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_LDS SQ_ACTIVE_INST_ANY --output-format csv -d $out/sq -o sq -- python3 bench.py $P > /dev/null 2> $out/sq.err
# keep the merge small: the per-dispatch counter tables are large
python - $out <<'PY'
import csv, collections, json, sys, glob, os
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for f in glob.glob(os.path.join(out, "sq", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("hefx::", "")
        if not k.startswith("ks_"): continue
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == "SQ_INSTS_VALU": n[k] += 1
json.dump({k: dict(launches=n[k], **v) for k, v in agg.items()}, open(os.path.join(out, "sq_counters.json"), "w"), indent=1)
PY
rm -rf $out/pf/*counter_collection.csv $out/pw/*counter_collection.csv $out/pf/*/*counter_collection.csv $out/pw/*/*counter_collection.csv $out/sq
find $out -name "*_agent_info.csv" -delete
python tools/batch_sweep.py C3 > $out/batch_sweep_C3.json 2> $out/batch_sweep.err
ls $out
