"""CPU (-m "not gpu"): the tile kernel's depth keys carry the low bits of a near fragment's float z so that the winner's
gl_FragCoord.z -- finer than the 24-bit depth below window z 0.5 -- is recovered without rasterising anything twice.
tests/near_key_check.c restates the two device functions involved (z24_of, near_z_from_key) and checks the round trip for
EVERY float in the range the encoding claims, for every key shift the library can choose (763 M cases, ~4 s)."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))


def test_every_near_float_is_recovered_from_its_key(tmp_path):
    exe = str(tmp_path / "near_key_check")
    subprocess.check_call(["gcc", "-O2", "-ffp-contract=off", "-o", exe, os.path.join(HERE, "near_key_check.c"), "-lm"])
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and r.stdout.startswith("ok "), r.stdout
    assert int(r.stdout.split()[1]) > 700_000_000


def test_the_check_mirrors_the_device_code():
    """The constants and the arithmetic of the C mirror are the device code's (a change on one side must show up here)."""
    dev = open(os.path.join(HERE, "..", "realtime_urdf_filter_amd", "csrc", "rtuf_kernels.hip")).read()
    chk = open(os.path.join(HERE, "near_key_check.c")).read()
    for token in ("5.9604648328104515e-08f", "16777215.0f", "cand = d > half ? cand - span : (d < -half ? cand + span : cand);",
                  "uint32_t cand = (cb & ~(span - 1u)) | low;"):
        assert token in dev and token in chk, token
    assert "kf.zexact = near_tile ? 1u << (26 - a.key_shift) : 8388609u;" in dev
    assert "c->key_shift = std::min(32 - order_bits, 16);" in open(os.path.join(HERE, "..", "realtime_urdf_filter_amd", "csrc", "rtuf_api.cpp")).read()
