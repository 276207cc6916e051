"""CPU: the C-ABI library loads and exports every symbol include/rtuf.h declares; without a GPU
it refuses to create a context (there is no CPU fallback)."""
import ctypes
import os
import re

import pytest

import realtime_urdf_filter_amd as R
from realtime_urdf_filter_amd import _capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "rtuf.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(rtuf_[a-z_0-9]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    lib = R.load_library()
    syms = declared_symbols()
    assert len(syms) >= 25
    for s in syms:
        assert hasattr(lib, s), "librtuf.so does not export %s" % s
    assert set(_capi.SYMBOLS) == set(syms), "python binding list out of date: %r" % (set(_capi.SYMBOLS) ^ set(syms))


def test_abi_version_and_struct_sizes():
    lib = R.load_library()
    assert lib.rtuf_abi_version() == R.ABI_VERSION == 6
    assert ctypes.sizeof(_capi.Params) == 48 and ctypes.sizeof(_capi.Stats) == 248
    p = R.default_params()
    assert abs(p.near_plane - 0.1) < 1e-7 and p.far_plane == 8.0 and abs(p.depth_distance_threshold - 0.05) < 1e-7
    assert p.filter_replace_value == 0.0 and p.flags == 0


def test_projection_from_intrinsics_matches_reference_formula():
    P, tx, ty = R.projection_from_intrinsics(525.0, 526.0, 319.5, 239.5, 640, 480, 0.1, 8.0, Tx=-39.4, Ty=0.0)
    assert P[0] == -2.0 * 525.0 / 640 and P[5] == 2.0 * 526.0 / 480
    assert P[8] == 2.0 * (0.5 - 319.5 / 640) and P[9] == 2.0 * (239.5 / 480 - 0.5)
    assert P[10] == -(8.0 + 0.1) / (8.0 - 0.1) and P[14] == -2.0 * 8.0 * 0.1 / (8.0 - 0.1) and P[11] == -1
    assert tx == -1 * (-39.4 / 525.0) and ty == 0.0
    assert all(P[i] == 0 for i in (1, 2, 3, 4, 6, 7, 12, 13, 15))


def _has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


@pytest.mark.skipif(_has_gpu(), reason="only meaningful on a machine without a GPU")
def test_no_cpu_fallback():
    with pytest.raises(R.RtufError) as e:
        R.Context(64, 48)
    assert e.value.code == _capi.RTUF_ERR_NO_DEVICE
    assert "no CPU fallback" in str(e.value)


def test_product_package_never_touches_the_oracle():
    """The oracle is test infrastructure: nothing under realtime_urdf_filter_amd/ may import it."""
    pkg = os.path.join(ROOT, "realtime_urdf_filter_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h", ".hpp", ".sh")):
                txt = open(os.path.join(dp, f), errors="replace").read()
                assert "oracle" not in txt.replace("the oracle", "").replace("oracle/", "").lower() or "import oracle" not in txt, f
                assert "from oracle" not in txt and "import oracle" not in txt and "librtuf_oracle" not in txt, f
                # ... and nothing of the benchmark's world (robot generator, workloads, per-GPU shares) ships in it
                assert "bench_support" not in txt, f


def test_product_package_has_no_checker_helpers():
    """No module, class or function of the product package is named after the oracle (helpers that prepare the
    checker's inputs live in bench_support/), and the package does not contain the benchmark's modules."""
    import importlib
    import inspect
    import pkgutil
    import realtime_urdf_filter_amd as R
    names = [m.name for m in pkgutil.iter_modules(R.__path__)]
    assert not {"synthetic", "workloads", "configs"} & set(names), names
    for name in names:
        mod = importlib.import_module("realtime_urdf_filter_amd." + name)
        for attr, obj in vars(mod).items():
            assert "oracle" not in attr.lower(), (name, attr)
            if inspect.isclass(obj) and obj.__module__ == mod.__name__:
                for meth in vars(obj):
                    assert "oracle" not in meth.lower(), (name, attr, meth)


def test_one_hip_runtime_per_process_in_either_import_order():
    """librtuf.so and a ROCm PyTorch wheel must end up on the same libamdhip64 whichever is used first
    (two runtimes in one process: the second one finds no GPU)."""
    import subprocess
    import sys
    code = r'''
import sys
order = sys.argv[1]
def first():
    import realtime_urdf_filter_amd as R
    R.load_library()
def second():
    import torch
for f in ((first, second) if order == "rtuf_first" else (second, first)):
    f()
libs = sorted({l.split()[-1] for l in open("/proc/self/maps") if "libamdhip64" in l})
print(len(libs), libs)
'''
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for order in ("rtuf_first", "torch_first"):
        out = subprocess.run([sys.executable, "-c", code, order], cwd=root, capture_output=True, text=True, timeout=300)
        assert out.returncode == 0, out.stderr[-2000:]
        assert out.stdout.split()[0] == "1", (order, out.stdout)


def test_expand_mask_bits_host_helper():
    """rtuf_expand_mask_bits (no GPU): select(bit, replace, sensor) for 32FC1; for 16UC1 the reference's two
    convertTo roundings, i.e. exactly what filter.py's host conversions give."""
    import numpy as np
    from realtime_urdf_filter_amd.filter import depth_f32_to_u16, depth_u16_to_f32
    rng = np.random.default_rng(1)
    for W, H in ((64, 5), (100, 7), (33, 3)):
        rw = (W + 31) // 32
        flags = rng.random((H, W)) < 0.4
        packed = np.zeros((H, rw * 32), np.uint8)
        packed[:, :W] = flags
        bits = np.packbits(packed, axis=1, bitorder="little").view(np.uint32).reshape(-1)
        assert bits.size == R.load_library().rtuf_mask_bits_words(W, H)
        d = rng.uniform(0.3, 9.0, (H, W)).astype(np.float32)
        d[0, 0], d[0, 1], d[1, 0] = np.nan, np.inf, 0.0
        masked, mask = R.expand_mask_bits(d, bits, 5.0)
        assert np.array_equal(mask, np.where(flags, 255, 0).astype(np.uint8))
        assert np.array_equal(masked.view(np.uint32), np.where(flags, np.float32(5.0), d).view(np.uint32))
        u = rng.integers(0, 65536, (H, W)).astype(np.uint16)
        masked16, mask16 = R.expand_mask_bits(u, bits, 5.0)
        expect = depth_f32_to_u16(np.where(flags, np.float32(5.0), depth_u16_to_f32(u)))
        assert np.array_equal(masked16, expect) and np.array_equal(mask16, mask)
        only_mask = R.expand_mask_bits(d, bits, 5.0, want_masked=False)
        assert only_mask[0] is None and np.array_equal(only_mask[1], mask)
