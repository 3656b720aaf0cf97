"""GPU: each hand-written HIP kernel (called through the C ABI debug entry points) against a plain
fp32 reference of the same op.  Tolerances: f16x3 (2-term split, the default) must be fp32-class
(1e-5); f16 (single product == TF32-class mantissa) 2e-3."""
import pytest

pytestmark = pytest.mark.gpu

# head_mx: the DPT head's arithmetic in the default policy (correction products as one block-scaled fp8 MFMA, ~1e-5 per GEMM),
# for the kernels the head is made of
TOL = {"f16x3": 2e-5, "f16": 3e-3, "head_mx": 6e-5, "mlp_mx": 6e-5}
PRECS = ["f16x3", "f16"]
HEAD_PRECS = ["f16x3", "f16", "head_mx"]


@pytest.fixture(scope="module")
def G():
    import gpu_checks
    return gpu_checks


@pytest.mark.parametrize("prec", HEAD_PRECS)
@pytest.mark.parametrize("kw", [dict(), dict(resid=True), dict(M=520, N=384, K=1024), dict(act=1, via_f16=1),
                                dict(act=2, via_f16=1), dict(M=1, N=96, K=32), dict(M=129, N=129, K=64),
                                # 256-row direct-to-LDS family (forced on small shapes): 256x256 and 256x128 tiles, M tails
                                dict(M=700, N=512, K=256, variant=2), dict(M=300, N=384, K=96, variant=2, resid=True),
                                dict(M=513, N=256, K=1024, variant=2, act=1, via_f16=1), dict(M=5, N=128, K=32, variant=2, act=2, via_f16=1),
                                # 192-row tiles (fractional DMA slot assignment)
                                dict(M=700, N=512, K=256, variant=3), dict(M=385, N=384, K=96, variant=3, resid=True),
                                dict(M=193, N=128, K=64, variant=3, act=1, via_f16=1)])
def test_gemm(G, prec, kw):
    r = G.check_gemm(prec, **kw)
    assert r["rel_l2"] < TOL[prec], r


@pytest.mark.parametrize("prec", PRECS)
@pytest.mark.parametrize("kw", [dict(), dict(hp=14, wp=14, pose_tok=0, S=1), dict(hp=3, wp=5, pose_tok=1, S=3),
                                dict(hp=14, wp=14, pose_tok=1, S=2, variant=2), dict(hp=5, wp=7, pose_tok=0, S=3, Cdim=256, K=256, variant=2),
                                dict(hp=14, wp=14, pose_tok=1, S=2, variant=3),
                                # throughput tile (192x128: >= 192 tiles), token rows 32-aligned: the V^T LDS-transpose path; with a pose
                                # token in front (t0 = 32k + 1: unaligned) its fallback
                                dict(hp=24, wp=32, pose_tok=0, S=4, K=64, Cdim=512), dict(hp=24, wp=32, pose_tok=1, S=4, K=64, Cdim=512)])
def test_qkv_rope(G, prec, kw):
    r = G.check_qkv_rope(prec, **kw)
    assert r["q"] < TOL[prec] and r["k"] < TOL[prec] and r["v"] < TOL[prec], r
    assert r["vpad_abs"] == 0.0, "V^T padding must stay zero (0 * garbage = NaN otherwise)"


@pytest.mark.parametrize("kw", [dict(), dict(resid=True), dict(M=520, N=384, K=1024), dict(M=520, N=384, K=1024, resid=True), dict(act=1, via_f16=1),
                                dict(M=129, N=128, K=64, resid=True), dict(M=700, N=512, K=256, variant=3), dict(M=385, N=384, K=96, variant=3, resid=True),
                                dict(M=700, N=512, K=256, variant=2, resid=True), dict(M=193, N=128, K=64, variant=3, act=1, via_f16=1),
                                dict(M=3000, N=1024, K=512, resid=True), dict(M=2400, N=768, K=3072, resid=True), dict(M=3000, N=512, K=256, act=1, via_f16=1)])
def test_gemm_mlp_f16mx(G, kw):
    """Precision f16x3m (round 6): mlp.fc2 in the f16mx arithmetic - fp32 and in-place-residual epilogues on f16mx rows / weights on
    the small-grid, 192x128 and 192x256 families - and mlp.fc1's GELU epilogue writing the f16mx rows (LDS-staged and edge tiles)."""
    r = G.check_gemm("mlp_mx", **kw)
    assert r["rel_l2"] < TOL["mlp_mx"], r


@pytest.mark.parametrize("kw", [dict(resid=True), dict(variant=3, N=768, K=1024, tiles_m=22, resid=True), dict(tail=1), dict(tail=32, resid=True)])
def test_gemm_tail_rows_mlp_f16mx(G, kw):
    """The pose-token tail blocks of mlp.fc2 in the f16mx arithmetic (gemm2_tail<MX>: fragments straight from global memory)."""
    r = G.check_gemm_tail("mlp_mx", **kw)
    assert r["rel_l2"] < TOL["mlp_mx"], r


@pytest.mark.parametrize("prec", ["f16x3", "f16"])
@pytest.mark.parametrize("kw", [dict(), dict(resid=True), dict(act=1, via_f16=1), dict(variant=3, N=768, K=1024, tiles_m=22, resid=True),
                                dict(variant=2, N=4096, K=128, tiles_m=6, act=1, via_f16=1), dict(tail=1), dict(tail=32, resid=True)])
def test_gemm_tail_rows(G, prec, kw):
    """Skinny tail blocks (the decoder's pose-token rows) on every family / epilogue the decoder uses them with."""
    r = G.check_gemm_tail(prec, **kw)
    assert r["rel_l2"] < TOL[prec], r


@pytest.mark.parametrize("prec", PRECS)
@pytest.mark.parametrize("kw", [dict(), dict(hp=14, wp=14, S=2), dict(hp=14, wp=14, S=2, variant=3), dict(hp=24, wp=32, S=4, K=64),
                                dict(hp=24, wp=32, S=4, K=64, Cdim=512)])      # 192x128 tiles + tail blocks: V^T LDS-transpose path
def test_qkv_rope_decoder_rows(G, prec, kw):
    r = G.check_qkv_rope_decoder_rows(prec, **kw)
    assert r["q"] < TOL[prec] and r["k"] < TOL[prec] and r["v"] < TOL[prec], r
    assert r["vpad_abs"] == 0.0, r


@pytest.mark.parametrize("prec", PRECS)
@pytest.mark.parametrize("kw", [dict(), dict(kv_shift=1), dict(n=768, S=2, heads=1, sharp=3.0), dict(n=12, sharp=6.0), dict(n=64),
                                dict(n=129, sharp=10.0, S=3, heads=1, kv_shift=2), dict(n=128, S=3), dict(n=127, sharp=3.0), dict(n=256, S=1, heads=3),
                                dict(n=255, S=1, heads=1)])
def test_attention_pose_token(G, prec, kw):
    r = G.check_attention_pose(prec, **kw)
    assert r["nan"] == 0, r
    assert r["rel_l2"] < TOL[prec] and r["rel_l2_pose"] < TOL[prec], r


@pytest.mark.parametrize("prec", PRECS)
@pytest.mark.parametrize("kw", [dict(), dict(kv_shift=1), dict(nq=70, nk=130, sharp=6.0),
                                dict(S=1, heads=1, nq=769, nk=769, sharp=3.0), dict(nq=1, nk=1), dict(nq=64, nk=64),
                                dict(nq=65, nk=129, sharp=10.0)])
def test_attention(G, prec, kw):
    r = G.check_attention(prec, **kw)
    assert r["nan"] == 0, r
    assert r["rel_l2"] < TOL[prec], r


@pytest.mark.parametrize("prec", HEAD_PRECS)
@pytest.mark.parametrize("kw", [dict(), dict(stride=2), dict(stride=2, H=6, W_=8),
                                dict(relu_in=1, act=2, resid=True, Cin=96, Co=256, H=9, W_=12), dict(H=1, W_=1),
                                dict(relu_in=1, act=2, resid=True, Cin=96, Co=256, H=19, W_=23, variant=2),
                                dict(Cin=64, Co=128, H=17, W_=9, variant=2), dict(stride=2, Cin=32, Co=256, H=15, W_=14, variant=2),
                                dict(relu_in=1, act=2, resid=True, Cin=96, Co=256, H=19, W_=23, variant=3), dict(Cin=64, Co=128, H=17, W_=9, variant=3),
                                # tiny grids with a long K loop: split-K partial sums + splitk_finish_kernel (SLAM-scale DPT levels)
                                dict(Cin=128, Co=64, H=6, W_=6), dict(relu_in=1, act=2, resid=True, Cin=256, Co=256, H=7, W_=7, n=3),
                                dict(stride=2, Cin=768, Co=256, H=14, W_=14, n=1)])
def test_conv3x3(G, prec, kw):
    r = G.check_conv3(prec, **kw)
    assert r["rel_l2"] < TOL[prec], r


@pytest.mark.parametrize("prec", HEAD_PRECS)
@pytest.mark.parametrize("kw", [dict(Cin=32, Co=128, H=9, W_=40), dict(Cin=96, Co=256, H=19, W_=23, relu_in=1, act=2, resid=True),
                                dict(Cin=64, Co=128, H=8, W_=32, n=3), dict(Cin=128, Co=256, H=17, W_=64, n=1, relu_in=1),
                                dict(Cin=256, Co=128, H=3, W_=97, act=2), dict(Cin=32, Co=256, H=1, W_=1), dict(Cin=64, Co=128, H=30, W_=33, resid=True)])
def test_conv3x3_halo_tiles(G, prec, kw):
    """conv3h.h (forced with tile family 8): pixel tiles of 8 x 32 outputs, halo in LDS, image borders / ragged tiles /
    several channel blocks (double-buffered halo) / ReLU on fragments / residual planes."""
    r = G.check_conv3(prec, variant=8, **kw)
    assert r["rel_l2"] < TOL[prec], r


@pytest.mark.parametrize("prec", HEAD_PRECS)
@pytest.mark.parametrize("kw", [dict(), dict(Cdim=192, k=2), dict(Cdim=192, k=2, variant=2), dict(Cdim=96, k=4, variant=2, H=9, W_=11)])
def test_convt(G, prec, kw):
    r = G.check_convt(prec, **kw)
    assert r["rel_l2"] < TOL[prec], r


@pytest.mark.parametrize("prec", HEAD_PRECS)
@pytest.mark.parametrize("kw", [dict(), dict(H=2, W_=3, crop=(3, 5)), dict(H=1, W_=1)])
def test_up2(G, prec, kw):
    r = G.check_up2(prec, **kw)
    assert r["rel_l2"] < TOL[prec], r


@pytest.mark.parametrize("prec", PRECS)
@pytest.mark.parametrize("C", [128, 768, 1024])
def test_layernorm(G, prec, C):
    r = G.check_layernorm(prec, Cdim=C)
    assert r["f32"] < 1e-5, r
    assert r["planes"] < TOL[prec], r


@pytest.mark.parametrize("prec", HEAD_PRECS)
def test_reference_op_goldens(G, prec):
    """Vectors produced by the reference's own modules (RoPE2D, svd_orthogonalize, postprocess...)."""
    r = G.check_ops_golden(prec)
    assert r["rope2d"] < 1e-5 and r["rope2d_roundtrip"] < 1e-5, r
    assert r["rope2d_f16"] < 2e-3 and r["rope2d_f64"] < 1e-5, r        # fp16 storage: one ulp of the stored half
    assert r["layernorm"] < 1e-5, r
    assert r["svd_orth"] < 1e-5, r
    assert r["bilinear"] < TOL[prec], r
    assert r["post_pts"] < TOL[prec] * 5 and r["post_conf"] < TOL[prec] * 5, r
