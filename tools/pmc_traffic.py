#!/usr/bin/env python3
"""Folds the two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs, as MI355X_MICROARCH.md prescribes)
of the bench command into profiles/<round>_bench_pmc_traffic.json: HBM bytes per kernel and per op.
usage: pmc_traffic.py <dir with FETCH_SIZE csv> <dir with WRITE_SIZE csv> <ops in the run> <out.json> [command text]
FETCH_SIZE / WRITE_SIZE count kilobytes; on gfx950 FETCH_SIZE under-counts 128-byte streaming requests by 2x
(MI355X_MICROARCH.md, HBM section), so both the raw and the doubled figure are kept."""
import csv, glob, json, os, sys
from collections import defaultdict


def fold(d, counter):
    per = defaultdict(lambda: [0, 0.0])
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if row["Counter_Name"] != counter:
                continue
            name = row["Kernel_Name"].split("(")[0].replace("void ", "").replace("hefx::", "")
            per[name][0] += 1
            per[name][1] += float(row["Counter_Value"])
    return per


def main():
    fdir, wdir, out = sys.argv[1], sys.argv[2], sys.argv[4]
    cmd = sys.argv[5] if len(sys.argv) > 5 else ""
    fe, wr = fold(fdir, "FETCH_SIZE"), fold(wdir, "WRITE_SIZE")
    # <ops in the run>: a number, or "per-chunk:<items>" = every ks_ntt_digits dispatch stands for one chunk of that many
    # items (robust against the number of passes the command makes: timed steps, warm steps, the profiled pass)
    if sys.argv[3].startswith("per-chunk:"):
        ops = int(sys.argv[3].split(":")[1]) * max(v[0] for k, v in fe.items() if k.startswith("ks_ntt_digits"))
    else:
        ops = int(sys.argv[3])
    rows, tf, tw = [], 0.0, 0.0
    for k in sorted(set(fe) | set(wr)):
        if not k.startswith(("ks_", "rs_")):
            continue
        rows.append({"kernel": k, "dispatches": fe.get(k, wr.get(k))[0], "FETCH_SIZE_KB": fe.get(k, [0, 0.0])[1],
                     "WRITE_SIZE_KB": wr.get(k, [0, 0.0])[1]})
        tf += fe.get(k, [0, 0.0])[1]
        tw += wr.get(k, [0, 0.0])[1]
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from seal_fyp_logistic_regression_amd import _build
    # the counters belong to the engine they were measured on: bench.py compares these hashes with the running library's
    res = {"command": cmd, "ops": ops, "csrc_sha16": _build.source_sha16(), "libhefx_sha16": _build.library_sha16(),
           "per_kernel": rows,
           "per_op_bytes": {"fetch_raw": tf * 1024 / ops, "fetch_x2": 2 * tf * 1024 / ops, "write": tw * 1024 / ops}}
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res["per_op_bytes"]))


if __name__ == "__main__":
    main()
