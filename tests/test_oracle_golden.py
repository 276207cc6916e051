"""CPU: the oracle (oracle/rtuf_oracle.c) reproduces every golden fixture bit for bit.  The
fixtures are outputs of the reference's GLSL shaders executed on Mesa llvmpipe."""
import numpy as np
import pytest

import golden_io
from oracle import bindings as O


@pytest.mark.parametrize("name", golden_io.fixture_names())
def test_oracle_reproduces_reference_output(name):
    fx = golden_io.Fixture(name)
    assert "llvmpipe" in fx.renderer
    masked, mask = O.filter_frame(fx.depth, fx.projection, fx.draws, fx.offset_inv, fx.cam_tf,
                                  max_diff=fx.max_diff, replace_value=fx.replace_value)
    fx.check(masked, mask)
    assert np.array_equal(masked.view(np.uint32), fx.expected_masked().view(np.uint32))


def test_there_are_fixtures():
    assert len(golden_io.fixture_names()) >= 10


@pytest.mark.parametrize("variant", ["vs_fma", "interp_fma", "cw_swap_12", "edge_rule_flip", "clip_old_t", "frag_div_rcp", "vp_fma"])
def test_numerical_variants_are_pinned(variant):
    """Every numerical choice of the oracle is forced by the reference's output: flipping any one of
    them breaks the z plane somewhere.  z differences surface in the mask only when the sensor value
    sits on the threshold, so this uses the debug z of the default variant as the yardstick."""
    fx = golden_io.Fixture("soup_seed12_160x120")
    base = O.filter_frame(fx.depth, fx.projection, fx.draws, fx.offset_inv, fx.cam_tf, want_debug=True)
    default = {"vs_fma": 0, "vp_fma": 1, "clip_vp_fma": 0, "interp_fma": 1, "frag_div_rcp": 0, "cw_swap_12": 0,
               "edge_rule_flip": 0, "clip_old_t": 0}
    try:
        O.set_variants(**{variant: 1 - default[variant]})
        alt = O.filter_frame(fx.depth, fx.projection, fx.draws, fx.offset_inv, fx.cam_tf, want_debug=True)
    finally:
        O.set_variants(**default)
    if variant == "edge_rule_flip":
        # only matters for exactly horizontal edges through pixel centres: the analytic fixture has them
        fx2 = golden_io.Fixture("analytic_edges_160x120")
        try:
            O.set_variants(edge_rule_flip=1)
            _, mask = O.filter_frame(fx2.depth, fx2.projection, fx2.draws, fx2.offset_inv, fx2.cam_tf,
                                     max_diff=fx2.max_diff, replace_value=fx2.replace_value)
        finally:
            O.set_variants(**default)
        assert (mask != fx2.mask).sum() > 0
        return
    if variant == "frag_div_rcp":
        # changes the shader arithmetic, not z: visible on the threshold fixture's mask
        fx2 = golden_io.Fixture("threshold_ulps_160x120")
        try:
            O.set_variants(frag_div_rcp=1)
            _, mask = O.filter_frame(fx2.depth, fx2.projection, fx2.draws, fx2.offset_inv, fx2.cam_tf,
                                     max_diff=fx2.max_diff, replace_value=fx2.replace_value)
        finally:
            O.set_variants(**default)
        assert (mask != fx2.mask).sum() > 0
        return
    differs = (base[2].view(np.uint32) != alt[2].view(np.uint32)).sum() + (base[1] != alt[1]).sum()
    assert differs > 0, "variant %s does not change anything on this fixture" % variant
