"""CPU (-m "not gpu"): the bit-pattern identities of the round-6 tile kernel.

gfx950 issues float add / mul / fma, integer add / sub, logic and right shifts at two cycles per wave64 instruction and
conversions, v_rndne, compares, selects, 24-bit multiplies and left shifts at four (profiles/valu_peak.json); the hot walks
replace instructions of the second kind by ones of the first wherever both give the same bits.  Whether
they do is arithmetic, not a GPU matter: tests/fast_class_check.c restates the device functions and checks them for every
float z > 0.5, every 24-bit depth of the upper half and 120 M vertex pairs (~15 s)."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))


def test_bit_pattern_identities_hold_for_every_input(tmp_path):
    exe = str(tmp_path / "fast_class_check")
    subprocess.check_call(["gcc", "-O2", "-ffp-contract=off", "-o", exe, os.path.join(HERE, "fast_class_check.c"), "-lm"])
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and r.stdout.startswith("ok "), r.stdout
    assert int(r.stdout.split()[1]) > 1_100_000_000


def test_the_check_mirrors_the_device_code():
    dev = open(os.path.join(HERE, "..", "realtime_urdf_filter_amd", "csrc", "rtuf_kernels.hip")).read()
    chk = open(os.path.join(HERE, "fast_class_check.c")).read()
    for token in ("0x4A800000u", "0x3E800001u", "16777215.0f",
                  "((uint32_t)(dcdx + dcdx) - ((uint32_t)(0 - dcdy) >> 31)) >> 31",
                  "const int X = xs[i] >> 8, Y = ys[i] >> 8, xf = xs[i] & 255, yf = ys[i] & 255;",
                  "(uint32_t)__mul24(dcdx, X) - (uint32_t)__mul24(dcdy, Y) + (uint32_t)(-((-t) >> 8))",
                  "const bool odd_w = ((lx1 - lx0) & 1) == 0;"):
        assert token in dev, token
    for token in ("0x4A800000u", "0x3E800001u", "16777215.0f", "((uint32_t)(dcdx + dcdx) - ((uint32_t)(0 - dcdy) >> 31)) >> 31"):
        assert token in chk, token
    # the identities are used only where their precondition holds: depths of the upper half in tiles without near geometry
    assert "(RTUF_FAST_CLASS && !LOW && MODE == 0) ? z24_of_upper_half(z) : z24_of(z)" in dev
    assert "if (RTUF_FAST_RESOLVE && !near_tile) {" in dev
    assert "return !(plane_min(a0, dzdx, dzdy, bx0, bx1, by0, by1) >= 0.51f) ? kNearBit : 0u;" in dev
