"""Every compile-time knob the tree still documents must keep compiling: the front end of hipcc
(-fsyntax-only, device pass, every template instantiated; no code generation, so seconds per knob instead of minutes)
over hefx_keyswitch.hip with each -DHEFX_* define.  The knobs whose experiments are closed were deleted in round 4
(HEFX_EPI*, HEFX_MAC_WAVES, HEFX_MAC_GROUP, HEFX_NB_FWD, HEFX_EO_LANE_MIN, HEFX_INV_NAT): profiles/EXPERIMENTS.md."""
import os
import re
import shutil
import subprocess
from concurrent.futures import ThreadPoolExecutor

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "seal_fyp_logistic_regression_amd", "csrc")
# knob -> what it is for (the documentation the test keeps honest)
KNOBS = {
    "-DHEFX_ONLY_LOGN=14": "development builds that instantiate one ring size (tools/build_variant.sh, tools/probe_ks.sh)",
    "-DHEFX_ONLY_LOGN=13": "... any ring size",
    "-DHEFX_STAMP=1": "clock stamps inside the small-batch kernels (tools/stamp_timeline.py)",
    "-DHEFX_WAVES=2": "one register budget for every NTT kernel (occupancy experiments)",
    "-DHEFX_SMALL_MAX=8": "how many descriptors travel in the first launch's kernel arguments",
    "-DHEFX_NO_L16": "A/B: the [0,8q) integer butterfly everywhere (round 3's 16q range off)",
    "-DHEFX_NO_LT2Q": "A/B: canonical MAC operands and results (round 3's lazy words off)",
    "-DHEFX_NO_EO_LANE": "A/B: column t in the forward transforms' first reads (round 3's lane-contiguous loader off)",
}


def _hipcc():
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", shutil.which("hipcc")):
        if cand and os.path.exists(cand):
            return cand
    return None


def test_documented_knobs_are_exactly_the_ones_in_the_sources():
    """a knob in the sources that this file does not list (or the other way round) fails: no silent growth"""
    found = set()
    for f in os.listdir(CSRC):
        if f.endswith((".hip", ".cuh", ".h", ".cpp")):
            for m in re.finditer(r"#\s*if(?:n?def|)\s+(?:!?\s*defined\s*\(\s*)?(HEFX_[A-Z0-9_]+)", open(os.path.join(CSRC, f)).read()):
                found.add(m.group(1))
    internal = {"HEFX_STAGE_FENCE", "HEFX_H", "HEFX_GLOBAL_AS"}  # macros, not knobs
    listed = {re.match(r"-D(HEFX_[A-Z0-9_]+)", k).group(1) for k in KNOBS}
    assert found - internal == listed, (sorted(found - internal - listed), sorted(listed - found))


def test_every_documented_knob_still_compiles():
    hipcc = _hipcc()
    if not hipcc:
        pytest.skip("no hipcc in this environment")
    src = os.path.join(CSRC, "hefx_keyswitch.hip")

    def one(flag):
        extra = [] if "ONLY_LOGN" in flag else ["-DHEFX_ONLY_LOGN=14"]  # one ring size is enough for a syntax pass
        cmd = [hipcc, "--offload-arch=gfx950", "--cuda-device-only", "-fsyntax-only", "-std=c++17", "-x", "hip", flag, *extra, src]
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
        return flag, r.returncode, r.stderr[-1500:]

    with ThreadPoolExecutor(max_workers=4) as ex:
        results = list(ex.map(one, KNOBS))
    bad = [(f, err) for f, rc, err in results if rc != 0]
    assert not bad, bad
