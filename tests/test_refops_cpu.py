"""CPU: pin the tests' own fp32 reference ops and the weight schema against the reference-derived
golden vectors (tests/golden/ops.npz was produced by oracle/gen_golden.py from the reference)."""
import numpy as np

from helpers import load_golden, rope2d_ref, max_rel
from vista_slam_amd import weights as W


def test_rope_ref_matches_reference_golden():
    g, _ = load_golden("ops")
    out = rope2d_ref(g["rope_tok"], g["rope_pos"], 100.0)
    assert max_rel(out, g["rope_out"]) < 2e-6


def test_schema_counts_match_reference():
    # 665 tensors / 438,455,505 distinct parameters (measured on the reference)
    sch = W.schema(W.FULL)
    assert len(sch) == 665
    seen, total = set(), 0
    for name, shape, _k, _f in sch:
        src = W._alias_of(name)
        if src in seen:
            continue
        seen.add(src)
        total += int(np.prod(shape))
    assert total == 438_455_505


def test_generator_is_deterministic_and_aliased():
    a = dict(W.generate(W.TINY, seed=43))
    b = dict(W.generate(W.TINY, seed=43))
    for k in ("init_pose_token", "dec_block.3.cross_attn.projk.weight"):
        assert np.array_equal(a[k], b[k])
    k1 = "downstream_head_pts.dpt.scratch.layer2_rn.weight"
    k2 = "downstream_head_pts.dpt.scratch.layer_rn.1.weight"
    assert np.array_equal(a[k1], a[k2])
    # known-answer: first elements of the hash stream (guards against platform drift)
    np.testing.assert_allclose(a["init_pose_token"].ravel()[:4],
                               [-0.03338519, -0.00919313, -0.01014909, 0.00090223], rtol=0, atol=1e-8)


def test_sharp_set_scales_only_qk():
    a = dict(W.generate(W.TINY, seed=43))
    b = dict(W.generate(W.TINY, seed=43, qk_gain=4.0))
    C = W.TINY.enc_embed_dim
    w0, w1 = a["enc_blocks.0.attn.qkv.weight"], b["enc_blocks.0.attn.qkv.weight"]
    np.testing.assert_allclose(w1[:2 * C], 4.0 * w0[:2 * C], rtol=1e-6)
    assert np.array_equal(w1[2 * C:], w0[2 * C:])
    assert np.array_equal(a["dec_block.0.cross_attn.projv.weight"], b["dec_block.0.cross_attn.projv.weight"])
