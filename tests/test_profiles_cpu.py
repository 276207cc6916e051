"""CPU: the OFFLINE figures bench.py quotes (roofline.traffic, valu_issue) must follow from the committed rocprofv3 counter text.

profiles/pmc_counters.json is what bench.py reads; scripts/pmc_to_json.py makes it from profiles/<tag>_pmc*.txt, which is what the
GPU box wrote.  A summary edited by hand, or one left behind by an older round's text, would put numbers in the driver's line
that no committed measurement supports."""
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PROFILES = os.path.join(ROOT, "profiles")


def test_counter_summary_follows_from_the_committed_counter_text(tmp_path):
    committed = json.load(open(os.path.join(PROFILES, "pmc_counters.json")))
    tag = re.search(r"profiles/(r\d\d)_pmc", committed["source"]).group(1)
    for name in os.listdir(PROFILES):          # the script reads and writes one directory: give it a copy of its inputs
        if name.startswith(tag + "_pmc") or name == "valu_peak_raw.json":
            (tmp_path / name).write_bytes(open(os.path.join(PROFILES, name), "rb").read())
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "pmc_to_json.py"), str(tmp_path), tag], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-1500:]
    again = json.load(open(tmp_path / "pmc_counters.json"))
    assert again["kernels"] == committed["kernels"] and again["c4_share"] == committed["c4_share"]
    tile = committed["kernels"]["tile_kernel<fused>"]
    # gfx950: FETCH_SIZE counts 128-byte requests at 64 (MI355X_MICROARCH.md): bytes = 2 x FETCH x 1024 + WRITE x 1024
    c = tile["counters_per_launch"]
    assert tile["hbm_bytes_per_launch"] == int(2 * c["FETCH_SIZE"] * 1024 + c["WRITE_SIZE"] * 1024)
    algorithmic = 9 * 640 * 480 * committed["streams_per_launch"]
    assert 1.0 <= tile["hbm_bytes_per_launch"] / algorithmic < 2.0
    # lanes live per issued VALU instruction (SQ_THREAD_CYCLES_VALU / (SQ_ACTIVE_INST_VALU x 64)): present for both big kernels
    for k in ("tile_kernel<fused>", "setup_kernel"):
        e = committed["kernels"][k]
        assert 0.3 < e["live_lane_fraction"] < 1.0
        assert abs(e["live_lane_fraction"] - e["counters_per_launch"]["SQ_THREAD_CYCLES_VALU"] / (e["counters_per_launch"]["SQ_ACTIVE_INST_VALU"] * 64.0)) < 1e-12


def test_profiles_index_is_what_the_committed_files_say():
    """profiles/INDEX.json maps every figure README.md / DESIGN.md quote to the file that backs it; its values are read out of
    those files by scripts/profiles_index.py, so a re-taken profile set without a re-made index (or an edited index) fails here."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "profiles_index.py"), "--check"], capture_output=True, text=True)
    assert r.returncode == 0, "profiles/INDEX.json is stale: run python scripts/profiles_index.py\n" + r.stderr[-800:]
    index = json.load(open(os.path.join(PROFILES, "INDEX.json")))
    assert len(index["entries"]) >= 30
    for e in index["entries"]:
        assert os.path.exists(os.path.join(ROOT, e["file"])), e["file"]
