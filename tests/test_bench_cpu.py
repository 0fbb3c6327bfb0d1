"""bench.py host logic that needs no GPU: the self-launcher refuses loudly when the devices are missing (VERDICT r1
item 1), the roofline helpers reproduce SURVEY 8(d)'s figures."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_gpus_n_without_devices_fails_loudly():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1"], env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0
    assert "needs 2 HIP devices" in r.stderr
    assert r.stdout.strip() == ""          # no JSON line from a run that did not happen


def test_one_rank_without_a_device_fails_loudly():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1"], env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "no CPU fallback" in (r.stderr + r.stdout)


def test_roofline_helpers_match_the_survey():
    sys.path.insert(0, ROOT)
    import bench
    assert bench.algorithmic_bytes_per_op(16384, 5) == 11141120
    assert bench.algorithmic_bytes_per_op(8192, 3) == 2555904        # C2
    assert bench.algorithmic_bytes_per_op(16384, 8) == 24117248      # C4
    N, primes = bench.SETS["C3"]
    v = bench.valu_bound_ops_per_s(N, primes, 5)
    assert v["int_transforms"] + v["f64_transforms"] == 6 * 7        # (L+1)(L+2) transforms per key switch
    assert (v["int_transforms"], v["f64_transforms"]) == (14, 28)
    assert 4.0e5 < v["peak_ops_per_s"] < 7.0e5
    N, primes = bench.SETS["C4"]
    v = bench.valu_bound_ops_per_s(N, primes, 8)
    assert v["int_transforms"] + v["f64_transforms"] == 9 * 10


def test_pmc_traffic_counts_ops_from_the_dispatches(tmp_path):
    """tools/pmc_traffic.py with `per-chunk:<items>`: the ops of a counter run are the digit-NTT dispatches times the
    chunk size, whatever number of passes (timed, warm, profiled) the bench command made -- a fixed op count once
    inflated the per-op traffic by 5/3 when the bench gained two warm steps."""
    import json
    hdr = "Kernel_Name,Counter_Name,Counter_Value\n"
    rows = lambda counter, kb, n: "".join(
        f'"void hefx::{k}(hefx::DevTables)",{counter},{kb}\n' for k in ("ks_ntt_digits_kernel<14>", "ks_mac_kernel<true>")
        for _ in range(n))
    for d, counter, kb in (("pf", "FETCH_SIZE", 1000.0), ("pw", "WRITE_SIZE", 500.0)):
        os.makedirs(tmp_path / d)
        (tmp_path / d / "x_counter_collection.csv").write_text(hdr + rows(counter, kb, 90))   # 5 passes of 18 chunks
    out = tmp_path / "t.json"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "pmc_traffic.py"), str(tmp_path / "pf"),
                        str(tmp_path / "pw"), "per-chunk:256", str(out), "synthetic"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    t = json.load(open(out))
    assert t["ops"] == 90 * 256
    assert abs(t["per_op_bytes"]["fetch_raw"] - 2 * 90 * 1000.0 * 1024 / (90 * 256)) < 1e-6
    assert abs(t["per_op_bytes"]["fetch_x2"] - 2 * t["per_op_bytes"]["fetch_raw"]) < 1e-6
    assert abs(t["per_op_bytes"]["write"] - 2 * 90 * 500.0 * 1024 / (90 * 256)) < 1e-6


def test_committed_traffic_and_counter_files_agree_with_the_bench_line():
    """The committed round-3 records are mutually consistent: the bench line's measured-HBM figure is the traffic file's
    per-op bytes, and the VALU instructions per op in roofline.valu.issue are the counter file's."""
    import json
    prof = os.path.join(ROOT, "profiles")
    line = json.loads(open(os.path.join(prof, "r03_bench.json")).read().strip().splitlines()[-1])
    pb = json.load(open(os.path.join(prof, "r03_bench_pmc_traffic.json")))["per_op_bytes"]
    assert abs(line["roofline"]["hbm_measured"]["bytes_per_op"] - (pb["fetch_x2"] + pb["write"])) < 1.0
    sq = json.load(open(os.path.join(prof, "r03_bench_sq_counters.json")))
    per_op = sum(v["SQ_INSTS_VALU"] / (v["launches"] * 256) for v in sq.values())
    assert abs(line["roofline"]["valu"]["issue"]["valu_wave_instr_per_op"] - per_op) < 1.0
    assert 1.4e6 < per_op < 1.7e6 and 15e6 < pb["fetch_x2"] + pb["write"] < 18e6
    assert line["verified"] is True and len(line["verified_items"]) == 19
