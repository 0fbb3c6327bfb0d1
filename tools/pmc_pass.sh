#!/bin/bash
# One rocprofv3 counter pass over the bench command, summed per key-switch kernel (GPU box):
#   tools/pmc_pass.sh outdir COUNTER [COUNTER ...]
out=$1; shift; mkdir -p $out; export TMPDIR=/tmp
rocprofv3 --pmc "$@" --output-format csv -d $out/p -o p -- python3 bench.py --steps 2 --warmup 0 --cpu-seconds 0 --variant-keys 0 --stream-keys 0 --key-per-item 0 --lt-direct 0 --secondary= --sustain 0 --lt= > /dev/null 2> $out/p.err
python - $out <<'PY'
import csv, collections, json, sys, glob, os
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob(os.path.join(out, "p", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("hefx::", "")
        if k.startswith("ks_"): agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
for k, v in agg.items(): print(k, {c: "%.4g" % x for c, x in v.items()})
PY
rm -rf $out/p
