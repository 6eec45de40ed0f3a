"""The D-FINE oracle (oracle/dfine_oracle.py) against golden vectors produced by the transformers functions the
reference's D-FINE path calls (tests/golden/make_dfine_golden.py).  CPU only."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import dfine_oracle as orc  # noqa: E402

G = np.load(os.path.join(ROOT, "tests", "golden", "dfine_golden.npz"))
SHAPES = [tuple(int(v) for v in hw) for hw in G["shapes"]]


@pytest.mark.parametrize("tag", ["a", "b"])
@pytest.mark.parametrize("method", ["default", "discrete"])
def test_msda_matches_transformers(tag, method):
    y = orc.multi_scale_deformable_attention_v2(G["value"], SHAPES, G[f"loc_{tag}"], G[f"attn_{tag}"],
                                                [int(n) for n in G[f"pts_{tag}"]], method)
    ref = G[f"msda_{tag}_{method}"]
    assert y.shape == ref.shape
    # fp32 sums of 12 products in a different order: a few ulp of the largest term
    assert np.abs(y - ref).max() <= 2e-6 * max(1.0, np.abs(ref).max())


def test_weighting_function_matches_transformers():
    w = orc.weighting_function(32, 0.5, 4.0)
    assert w.shape == (33,) and w[16] == 0 and np.all(np.diff(w) > 0)
    np.testing.assert_allclose(w, G["project"], rtol=2e-6, atol=1e-7)


def test_integral_and_distance2bbox_match_transformers():
    d = orc.integral(G["dist"], G["project"])
    np.testing.assert_allclose(d, G["integral"], rtol=1e-5, atol=1e-6)
    b = orc.distance2bbox(G["points"], G["integral"], 4.0)
    ref = G["boxes"]
    fin = np.isfinite(ref)
    assert np.array_equal(np.isnan(b), np.isnan(ref)) and np.array_equal(np.isinf(b), np.isinf(ref))
    np.testing.assert_allclose(b[fin], ref[fin], rtol=1e-6, atol=1e-7)
    bc = np.clip(b, 0, 1)   # np.clip keeps NaN like torch.clamp
    refc = G["boxes_clamped"]
    assert np.array_equal(np.isnan(bc), np.isnan(refc))
    np.testing.assert_allclose(bc[~np.isnan(refc)], refc[~np.isnan(refc)], rtol=1e-6, atol=1e-7)


def test_attention_module_matches_transformers():
    """DFineMultiscaleDeformableAttention.forward (random init, 4-d reference points) vs the restatement fed with the
    module's own linear-layer outputs."""
    y = orc.deformable_attention_module(G["value"], SHAPES, G["mod_ref"], G["mod_offsets"], G["mod_logits"], [4, 4, 4],
                                        float(G["mod_offset_scale"]))
    ref = G["mod_out"]
    assert y.shape == ref.shape and np.abs(y - ref).max() <= 2e-6 * max(1.0, np.abs(ref).max())
