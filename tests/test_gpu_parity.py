"""GPU: end-to-end parity of the HIP STA forward (through the C ABI / STAFrontend shim) with
golden vectors generated from the reference PyTorch model (oracle/gen_golden.py).

north_star tolerance: 1e-3 relative (fp32) on pointmaps, confidence and pose.  Every case is run
in the default precision (f16x3); the single-product fp16 mode is reported against a looser bound."""
import pytest

pytestmark = pytest.mark.gpu

TOL = 1e-3
KEYS_OUT = ["main_pts3d", "main_conf", "main_pose", "main_pose_conf", "supp_pts3d", "supp_conf", "supp_pose",
            "supp_pose_conf", "split_pose_vs_golden", "split_pts_vs_golden"]


@pytest.fixture(scope="module")
def G():
    import gpu_checks
    yield gpu_checks
    gpu_checks.drop_models()


DEFAULT = "f16x3h"      # the shipped precision policy: f16x3 transformer + pose head, f16mx arithmetic in the DPT head


def ctol(prec):
    """Self-consistency bound between two valid schedules of the same forward (different tiles / split-K / streams).  f16x3
    differs by summation order only; in f16x3h a head conv that lands on a split-K slab or on the register-staged tile runs
    full f16x3 instead of f16mx, so two schedules also differ by the f16mx correction error (2^-14-class per element with the
    e5m2 activation bytes; kernel-level bound in test_gpu_kernels.TOL["head_mx"])."""
    return 2e-5 if prec == "f16x3" else 6e-5
STRESS = [f"tiny_48x80_sharp_s{sd}{sm}" for sd in (44, 45, 46, 47) for sm in ("_smooth", "")]


@pytest.mark.parametrize("prec", [DEFAULT, "f16x3"])
@pytest.mark.parametrize("case", ["tiny_32x32_b1", "tiny_48x64_b2", "tiny_48x64_b2_sharp", "tiny_48x80_smooth_sharp"])
def test_tiny_goldens_default_precision(G, case, prec):
    r = G.run_golden_case(case, prec)
    bad = {k: v for k, v in r.items() if v > TOL}
    assert not bad, bad
    assert G.last_range == (0, 0), f"a plane writer saturated on an ordinary golden: {G.last_range} (sta_range_report)"


OUTLIER = ["tiny_48x64_b2_outlier", "tiny_48x80_outlier_sharp", "full_224_b1_outlier"]


@pytest.mark.parametrize("prec", [DEFAULT, "f16x3"])
@pytest.mark.parametrize("case", OUTLIER)
def test_outlier_statistics_goldens(G, case, prec):
    """Trained-checkpoint-like RANGE statistics from the imported reference (weights.py `_outlier`):
    LayerNorm gains with a 3..10x tail, three massive-activation channels (+80 / -60 / +50) in both residual streams, DPT
    feature maps 300x larger (up to 6.5e3: inside the fp16 range, and inside the +-57344 of the e5m2 correction bytes the
    head's f16mx arithmetic keeps for ACTIVATIONS - with e4m3 bytes (+-448) these cases sat at 2.3e-4..3.3e-4 with 3e5..1.4e6
    saturated bytes; with e5m2 they are 3e-5..6e-5 with none).  The shipped policy must hold the 1e-3 bar and nothing may
    leave either range."""
    if case.startswith("full"):
        G.drop_models()
    r = G.run_golden_case(case, prec)
    bad = {k: v for k, v in r.items() if v > TOL}
    assert not bad, (bad, G.last_range)
    assert G.last_range == (0, 0), f"saturation inside the range the reference golden covers: {G.last_range}"


@pytest.mark.parametrize("prec", [DEFAULT, "f16x3"])
def test_range_overflow_is_reported_not_silent(G, prec):
    """DPT feature maps 3e4x larger (up to 5e5: past the fp16 planes' +-65504; the fp32 reference is unaffected): the planes
    saturate, so the result is NOT expected to match - but the library must say so (sta_range_report counts[0] > 0)."""
    G.run_golden_case("tiny_48x64_b1_overflow", prec)
    assert G.last_range[0] > 0, G.last_range
    m = G.model("tiny", 1.0, prec, 43, 2)
    assert m.range_report() == (0, 0), "the report resets"


def test_range_counters_are_per_handle(G):
    """The range counters live in the handle (round 4; a device-wide pair before): a forward that saturates on one handle
    (the overflow golden's weights) leaves the report of a second handle on the same GPU at (0, 0)."""
    import torch
    from vista_slam_amd import weights as W
    clean = G.model("tiny", 1.0, DEFAULT, 43, 0)
    dirty = G.model("tiny", 1.0, DEFAULT, 43, 2)
    clean.range_report(); dirty.range_report()                  # reset both
    imgs = torch.from_numpy(W.synth_images(2, 48, 64, seed=43, tag=1)).cuda()
    dirty.forward_pair(imgs[:1], imgs[1:])
    clean.forward_pair(imgs[:1], imgs[1:])
    assert clean.range_report() == (0, 0), "events of another handle leaked into this one"
    assert dirty.range_report()[0] > 0, "the saturating handle lost its own events"


@pytest.mark.parametrize("case", STRESS)
def test_stress_goldens_default_precision(G, case):
    """Eight more draws (weight / image seeds 44-47, smooth and noisy frames) of the sharpened tiny configuration: these
    sets amplify every rounding error ~100x (the fp32 oracle itself is at 1e-4 on them), so one of them is a noisy judge of
    a precision policy.  The default policy must hold all of them at the 1e-3 bar (measured: <= 3.7e-4, same as pure f16x3)."""
    r = G.run_golden_case(case, DEFAULT)
    bad = {k: v for k, v in r.items() if v > TOL}
    assert not bad, bad
    assert G.last_range == (0, 0), G.last_range


@pytest.mark.parametrize("case", ["tiny_48x64_b2", "tiny_48x80_smooth_sharp"])
def test_tiny_goldens_large_tile_kernel(G, case):
    """Same goldens with the large-tile GEMM families forced (2 = 256x256 / 16 waves, 3 = 192x256 / 12 waves: the
    bench-scale kernels of mlp.fc1 / mlp.fc2 and the refinement convolutions, f16x3 and f16mx forms; 8 = the halo-tiled
    3x3 convolution kernel incl. the fused DPT tail, conv3h.h)."""
    for variant in (2, 3, 8):
        r = G.run_golden_case(case, DEFAULT, variant=variant)
        bad = {k: v for k, v in r.items() if v > TOL}
        assert not bad, (variant, bad)


@pytest.mark.parametrize("case", ["tiny_32x32_b1", "tiny_48x64_b2_sharp"])
def test_tiny_goldens_f16(G, case):
    r = G.run_golden_case(case, "f16")
    bad = {k: v for k, v in r.items() if v > (0.25 if 'sharp' in case else 2e-2)}
    assert not bad, bad


@pytest.mark.parametrize("case", ["full_224_b1", "full_384x512_b1"])
def test_full_goldens_default_precision(G, case):
    """Full-size model (438 M parameters) at the SLAM resolution and at the benchmark resolution."""
    r = G.run_golden_case(case, DEFAULT)
    bad = {k: v for k, v in r.items() if v > TOL}
    assert not bad, bad
    assert G.last_range == (0, 0), G.last_range


@pytest.mark.parametrize("case", ["tiny_80x48_b2_portrait", "full_512x384_b1_portrait"])
def test_portrait_goldens_default_precision(G, case):
    """Portrait frames (H > W): tokenised row-major as they are (PatchEmbedDust3R) and every per-pixel output returned as
    the transposed view [B,W,H,..] the reference's head wrapper produces (utils/misc.py:60-61,81); the full-size case is
    the benchmark resolution turned by 90 degrees (what f3 emits for a portrait camera)."""
    import torch
    from helpers import load_golden
    from vista_slam_amd import weights as W
    r = G.run_golden_case(case, DEFAULT)
    bad = {k: v for k, v in r.items() if v > TOL}
    assert not bad, bad
    g, meta = load_golden(case)
    H, W_, B = int(meta["H"]), int(meta["W"]), int(meta["B"])
    assert H > W_
    m = G.model("tiny" if case.startswith("tiny") else "full", 1.0, DEFAULT, int(meta["seed"]))
    imgs = torch.from_numpy(W.synth_images(2 * B, H, W_, seed=int(meta["seed"]), tag=0)).cuda()
    main, supp = m.forward_pair(imgs[:B], imgs[B:])
    assert tuple(main["pts3d_pred"].shape) == (B, W_, H, 3) and tuple(supp["conf"].shape) == (B, W_, H)
    out = m({"main_view": {"img": imgs[:B]}, "neighbor_views": [{"img": imgs[B:]}], "loop_views": []})
    assert tuple(out["main_views"][0]["pts3d_pred"].shape) == (B, W_, H, 3)
    assert torch.equal(out["support_views"][0]["conf"], supp["conf"]) or \
        float((out["support_views"][0]["conf"] - supp["conf"]).abs().max() / supp["conf"].abs().max()) < 1e-5
    if case.startswith("full"):
        G.drop_models()


def test_full_sharp_attention_golden(G):
    """Peaky-attention weight set: a wrong RoPE/softmax cannot hide under the tolerance."""
    G.drop_models()
    r = G.run_golden_case("full_224_b1_sharp", DEFAULT)
    bad = {k: v for k, v in r.items() if v > TOL}
    assert not bad, bad


STRESS_FULL = ["full_384x512_b1_sharp", "full_384x512_b1_outlier", "full_224_b1_sharp_s44_smooth", "full_224_b1_sharp_s45"]


@pytest.mark.parametrize("case", STRESS_FULL)
def test_full_architecture_stress_goldens_headline_resolution(G, case):
    """Round 4: peaky attention (Q/K gain 3) and checkpoint-like range statistics on the FULL architecture - at the headline
    resolution 384x512 (nq = 768: the pose side blocks of the attention kernel, 12 full key tiles, the 192x128 / 192x256 /
    256x256 GEMM families and the fused DPT tail on the halo kernel are paths the 224x224 goldens never take; with default
    weights attention is near-uniform and a wrong softmax / RoPE would hide) and two more full-depth sharp seeds at
    the SLAM resolution (smooth + noisy frames).  Both shipped arithmetic policies at the 1e-3 bar, nothing saturated, and the
    integer positions bit-exact (run_golden_case: pos_a / pos_b)."""
    G.drop_models()
    for prec in (DEFAULT, "f16x3"):
        r = G.run_golden_case(case, prec)
        bad = {k: v for k, v in r.items() if v > TOL}
        assert not bad, (prec, bad)
        assert G.last_range == (0, 0), (prec, G.last_range)
    G.drop_models()


def test_integer_positions_bit_exact(G):
    """`_encode_image` also returns the int64 (y, x) patch positions (PositionGetter, sta_blocks.py:241-247): compared entry by
    entry with the reference's tensor (every golden holds pos_a / pos_b since round 4), landscape, portrait and batch > 1."""
    import numpy as np
    import torch
    from helpers import load_golden
    for case in ("tiny_32x32_b1", "tiny_48x64_b2", "tiny_80x48_b2_portrait", "tiny_48x80_smooth_sharp"):
        g, meta = load_golden(case)
        H, W_, B = int(meta["H"]), int(meta["W"]), int(meta["B"])
        m = G.model("tiny", 1.0, DEFAULT, 43)
        _f, pos = m._encode_image(torch.zeros(B, 3, H, W_, device="cuda"), None, normalize=False)
        assert pos.dtype == torch.int64 and tuple(pos.shape) == g["pos_a"].shape
        assert np.array_equal(pos.cpu().numpy(), g["pos_a"]) and np.array_equal(pos.cpu().numpy(), g["pos_b"])
        _f, pos8 = m.encode_u8hwc(torch.zeros(B, H, W_, 3, dtype=torch.uint8, device="cuda"))
        assert np.array_equal(pos8.cpu().numpy(), g["pos_a"])


def test_curope_compat_module_vs_reference_rope_golden(G):
    """vista_slam_amd.curope_compat.cuRoPE2D - the class the reference's import switch (pos_embed.py:106-108) would pick up -
    against the reference RoPE2D vectors of ops.npz (incl. position -1), on the strided q / k views the attention layer hands
    it (sta_blocks.py:132-137: (B,H,N,D) views of the (B,N,3,H,D) qkv tensor), forward and inverse rotation."""
    import numpy as np
    import torch
    import vista_slam_amd.curope_compat as cc
    from helpers import load_golden, max_rel
    g, _ = load_golden("ops")
    tok = torch.from_numpy(g["rope_tok"]).cuda()                 # (B,H,N,D)
    pos = torch.from_numpy(g["rope_pos"]).cuda()
    B, Hh, N, D = tok.shape
    qkv = torch.zeros(B, N, 3, Hh, D, device="cuda")
    qkv[:, :, 1] = tok.permute(0, 2, 1, 3)
    qkv_t = qkv.transpose(1, 3)                                   # (B,H,3,N,D) view, like sta_blocks.py:132
    k = qkv_t[:, :, 1]
    rope = cc.cuRoPE2D(freq=100.0)
    out = rope(k, pos)
    assert out.data_ptr() == k.data_ptr()                         # in place, returns its argument (curope2d.py:38-40)
    assert max_rel(out.cpu().numpy(), g["rope_out"]) < 2e-6
    assert float(qkv[:, :, 0].abs().max()) == 0.0 and float(qkv[:, :, 2].abs().max()) == 0.0     # q / v slots untouched
    # the inverse rotation (fwd = -F0: what the reference's backward applies to the gradient, curope2d.py:24-29) undoes the forward
    t = torch.from_numpy(g["rope_tok"]).cuda().permute(0, 2, 1, 3).contiguous()      # (B,N,H,D): the kernel's layout
    y = t.clone()
    cc.rope_2d(y, pos, 100.0, 1.0)
    assert max_rel(y.permute(0, 2, 1, 3).cpu().numpy(), g["rope_out"]) < 2e-6
    cc.rope_2d(y, pos, 100.0, -1.0)
    assert max_rel(y.cpu().numpy(), t.cpu().numpy()) < 5e-6                           # R^T (R t) = t
    # inference-only: a tensor that requires grad is refused loudly (no silent gradient drop)
    with pytest.raises(RuntimeError):
        rope(torch.zeros(1, 1, 2, 4, device="cuda", requires_grad=True), torch.zeros(1, 2, 2, dtype=torch.int64, device="cuda"))
    # the module-level function has the extension's signature (curope.cpp:49-65) and refuses CPU tensors loudly
    with pytest.raises(RuntimeError):
        cc.rope_2d(torch.zeros(1, 1, 1, 4), torch.zeros(1, 1, 2, dtype=torch.int64), 100.0, 1.0)


def test_random_shapes_and_batches_vs_oracle(G):
    """Beyond the fixed golden shapes: random (H, W, B) - odd token grids, ragged last tiles, square, wide and portrait
    frames, a single 16x16 token - against the oracle (itself pinned to the reference goldens) on the tiny configuration, default and opt-in precision."""
    import numpy as np
    import torch
    from helpers import rel_l2
    from oracle import sta_oracle as O
    from vista_slam_amd import weights as W
    rng = np.random.default_rng(5)
    sd = W.state_dict(W.TINY, seed=43)
    shapes = [(int(hp), int(rng.integers(hp, 8)), int(rng.integers(1, 4))) for hp in rng.integers(1, 6, size=5)]
    shapes += [(1, 1, 2), (1, 7, 1), (7, 1, 1), (5, 2, 3), (6, 4, 1)]      # one-token frames, one-row / one-column grids, portrait
    for it, (hp, wp, B) in enumerate(shapes):
        H, Wd = 16 * hp, 16 * wp
        imgs = (W.smooth_images if it % 2 else W.synth_images)(2 * B, H, Wd, seed=43, tag=30 + it)
        want = O.forward_pair(W.TINY, sd, imgs[:B], imgs[B:])
        for prec, tol in (("f16x3", 2e-5), (DEFAULT, 5e-5)):
            m = G.model("tiny", 1.0, prec)
            G.set_variant(m, 0)
            main, supp = m.forward_pair(torch.from_numpy(imgs[:B]).cuda(), torch.from_numpy(imgs[B:]).cuda())
            torch.cuda.synchronize()
            for got, ref in ((main, want["main"]), (supp, want["supp"])):
                for k, rk in (("pts3d_pred", "pts3d"), ("conf", "conf"), ("relative_pose", "pose"), ("relative_pose_conf", "pose_conf")):
                    e = rel_l2(got[k].cpu().numpy(), ref[rk])
                    assert e < tol, (H, Wd, B, prec, k, e)


def test_error_paths(G):
    """Same failure behaviour as the reference: H,W % 16 (patch_embed.py:20-21), strict state_dict."""
    import numpy as np
    import torch
    from vista_slam_amd import _lib
    from vista_slam_amd import weights as W
    from vista_slam_amd.sta_frontend import STAFrontend
    m = G.model("tiny", 1.0, "f16x3")
    with pytest.raises(AssertionError):
        m._encode_image(torch.zeros(1, 3, 30, 32, device="cuda:0"), None, normalize=False)
    with pytest.raises(AssertionError):
        m.forward_pair(torch.zeros(1, 3, 64, 40, device="cuda:0"), torch.zeros(1, 3, 64, 40, device="cuda:0"))
    fresh = STAFrontend(W.TINY, "cuda:0")
    with pytest.raises(_lib.StaError, match="unexpected key"):
        fresh._load_one("not.a.key", np.zeros(3, np.float32))
    with pytest.raises(_lib.StaError, match="size mismatch"):
        fresh._load_one("dec_norm.weight", np.zeros(7, np.float32))
    with pytest.raises(_lib.StaError, match="missing key"):
        fresh.load_state_dict({"dec_norm.weight": np.zeros(128, np.float32)})
    with pytest.raises(_lib.StaError, match="not finalized"):
        fresh._encode_image(torch.zeros(1, 3, 32, 32, device="cuda:0"), None, normalize=False)


# ---------------------------------------------------------------------------------------------------
# Full-size (BASELINE configs[1]: 512x384, batch 8) properties that need no oracle run: the oracle
# would take minutes per pair at this size, so parity at scale is checked through size-independent
# invariants of the algorithm plus the sub-sampled reference golden of pair 0.
@pytest.fixture(scope="module")
def full_b8(G):
    import torch
    from vista_slam_amd import weights as W
    G.drop_models()
    m = G.model("full", 1.0, DEFAULT)
    G.set_variant(m, 0)
    imgs = W.synth_images(16, 384, 512, seed=43, tag=0)
    a, b = torch.from_numpy(imgs[:8]).cuda(), torch.from_numpy(imgs[8:]).cuda()
    main, supp = m.forward_pair(a, b)
    torch.cuda.synchronize()
    return m, a, b, main, supp


def test_full_size_batch8_outputs_are_well_formed(full_b8):
    import torch
    m, a, b, main, supp = full_b8
    for o in (main, supp):
        assert o["pts3d_pred"].shape == (8, 384, 512, 3) and o["conf"].shape == (8, 384, 512)
        assert bool(torch.isfinite(o["pts3d_pred"]).all()) and bool(torch.isfinite(o["conf"]).all())
        assert float(o["conf"].min()) >= 1.0                                  # conf = 1 + exp(x)
        R = o["relative_pose"][:, :3, :3].double()
        eye = torch.eye(3, dtype=torch.float64, device=R.device)
        assert float((R @ R.transpose(1, 2) - eye).abs().max()) < 1e-5        # pp.mat2SE3(atol=1e-3) needs this (slam.py:166)
        assert float((torch.linalg.det(R) - 1).abs().max()) < 1e-5
        assert bool((o["relative_pose"][:, 3] == torch.tensor([0., 0., 0., 1.], device=R.device)).all())
        assert float(o["relative_pose_conf"].min()) > 0 and float(o["relative_pose_conf"].max()) < 1


def test_full_size_batch_rows_are_independent(full_b8):
    """Pairs are independent units: pair i of the batch-8 run == the same pair run alone (what the
    multi-GPU sharding relies on).  Not bit-exact only because tile scheduling changes summation order."""
    import torch
    from helpers import rel_l2
    m, a, b, main, supp = full_b8
    for i in (0, 5):
        m1, s1 = m.forward_pair(a[i:i + 1], b[i:i + 1])
        torch.cuda.synchronize()
        assert rel_l2(m1["pts3d_pred"].cpu().numpy(), main["pts3d_pred"][i:i + 1].cpu().numpy()) < ctol(DEFAULT)
        assert rel_l2(s1["conf"].cpu().numpy(), supp["conf"][i:i + 1].cpu().numpy()) < ctol(DEFAULT)
        assert rel_l2(m1["relative_pose"].cpu().numpy(), main["relative_pose"][i:i + 1].cpu().numpy()) < ctol(DEFAULT)


@pytest.mark.parametrize("cfg", [(2, 224, 224), (4, 224, 224), (8, 224, 224), (2, 384, 512), (4, 384, 512)])
def test_mid_size_batches_match_the_golden_pair(G, cfg):
    """Batch sizes between the SLAM regime and the benchmark (where the tile cost model switches families: small-grid split-K
    below 192 tiles, 192x128 / 192x256 / 256x256 above, profiles/r03_tile_table.txt): pair 0 of every batch is the committed
    reference golden's pair (full_224_b1 / full_384x512_b1 use images 0 and 1 of the same procedural set), every other pair
    must equal the same pair run alone."""
    import numpy as np
    import torch
    from helpers import load_golden, rel_l2
    from vista_slam_amd import weights as W
    B, H, Wd = cfg
    G.drop_models()
    m = G.model("full", 1.0, DEFAULT)
    G.set_variant(m, 0)
    g, meta = load_golden("full_224_b1" if H == 224 else "full_384x512_b1")
    assert int(meta["H"]) == H and int(meta["W"]) == Wd and int(meta["seed"]) == 43
    im = W.synth_images(2, H, Wd, seed=43, tag=0)
    extra = W.synth_images(2 * B, H, Wd, seed=43, tag=5)
    a = torch.from_numpy(np.concatenate([im[:1], extra[:B - 1]])).cuda()
    b = torch.from_numpy(np.concatenate([im[1:], extra[B:2 * B - 1]])).cuda()
    main, supp = m.forward_pair(a, b)
    torch.cuda.synchronize()
    sub = int(meta["sub"])
    assert rel_l2(main["pts3d_pred"][:1].cpu().numpy()[:, ::sub, ::sub], g["main_pts3d"]) < TOL
    assert rel_l2(supp["conf"][:1].cpu().numpy()[:, ::sub, ::sub], g["supp_conf"]) < TOL
    assert rel_l2(main["relative_pose"][:1].cpu().numpy(), g["main_pose"]) < TOL
    assert rel_l2(supp["relative_pose"][:1].cpu().numpy(), g["supp_pose"]) < TOL
    i = B - 1
    m1, s1 = m.forward_pair(a[i:i + 1], b[i:i + 1])
    torch.cuda.synchronize()
    assert rel_l2(m1["pts3d_pred"].cpu().numpy(), main["pts3d_pred"][i:i + 1].cpu().numpy()) < ctol(DEFAULT)
    assert rel_l2(s1["conf"].cpu().numpy(), supp["conf"][i:i + 1].cpu().numpy()) < ctol(DEFAULT)
    assert rel_l2(s1["relative_pose"].cpu().numpy(), supp["relative_pose"][i:i + 1].cpu().numpy()) < ctol(DEFAULT)


def test_full_size_view_swap_symmetry(full_b8):
    """The decoder weights are shared between the two sides (sta_model.py:231-235), so swapping the views
    swaps the outputs: forward(b, a).main == forward(a, b).support."""
    import torch
    from helpers import rel_l2
    m, a, b, main, supp = full_b8
    main2, supp2 = m.forward_pair(b[:2], a[:2])
    torch.cuda.synchronize()
    assert rel_l2(main2["pts3d_pred"].cpu().numpy(), supp["pts3d_pred"][:2].cpu().numpy()) < ctol(DEFAULT)
    assert rel_l2(supp2["relative_pose"].cpu().numpy(), main["relative_pose"][:2].cpu().numpy()) < ctol(DEFAULT)
    assert rel_l2(supp2["conf"].cpu().numpy(), main["conf"][:2].cpu().numpy()) < ctol(DEFAULT)


def test_full_size_pair0_matches_reference_golden(full_b8):
    """Pair 0 of the batch-8 bench workload uses the same procedural images as the committed 384x512
    golden (tag 0, images 0 and 8 differ from the B=1 golden's images 0 and 1) - so compare the B=1 golden
    inputs explicitly through the same batched code path instead."""
    import torch
    from helpers import load_golden, rel_l2
    from vista_slam_amd import weights as W
    m = full_b8[0]
    g, meta = load_golden("full_384x512_b1")
    imgs = W.synth_images(2, 384, 512, seed=43, tag=0)
    pad = W.synth_images(14, 384, 512, seed=43, tag=3)
    a = torch.from_numpy(np_concat(imgs[:1], pad[:7])).cuda()
    b = torch.from_numpy(np_concat(imgs[1:], pad[7:])).cuda()
    main, supp = m.forward_pair(a, b)
    torch.cuda.synchronize()
    sub = int(meta["sub"])
    assert rel_l2(main["pts3d_pred"][:1].cpu().numpy()[:, ::sub, ::sub], g["main_pts3d"]) < TOL
    assert rel_l2(supp["conf"][:1].cpu().numpy()[:, ::sub, ::sub], g["supp_conf"]) < TOL
    assert rel_l2(main["relative_pose"][:1].cpu().numpy(), g["main_pose"]) < TOL
    assert rel_l2(supp["relative_pose"][:1].cpu().numpy(), g["supp_pose"]) < TOL


def test_full_size_batch8_every_slot_matches_reference_golden(full_b8):
    """BASELINE configs[1] at its full batch: every one of the 8 batch slots holds one of the two DIFFERENT pairs of the
    reference golden `full_384x512_b2` (order 0,1,1,0,0,1,0,1), and all eight output tensors of every slot are compared
    with the reference's outputs for that pair - a cross-slot mix-up or a slot-dependent error cannot hide."""
    import numpy as np
    import torch
    from helpers import load_golden, rel_l2
    from vista_slam_amd import weights as W
    m = full_b8[0]
    g, meta = load_golden("full_384x512_b2")
    sub = int(meta["sub"])
    imgs = W.synth_images(4, 384, 512, seed=int(meta["seed"]), tag=0)
    order = [0, 1, 1, 0, 0, 1, 0, 1]
    a = torch.from_numpy(np.ascontiguousarray(imgs[:2][order])).cuda()
    b = torch.from_numpy(np.ascontiguousarray(imgs[2:][order])).cuda()
    main, supp = m.forward_pair(a, b)
    torch.cuda.synchronize()
    for slot, src in enumerate(order):
        for side, o in (("main", main), ("supp", supp)):
            e = {"pts3d": rel_l2(o["pts3d_pred"][slot].cpu().numpy()[::sub, ::sub], g[f"{side}_pts3d"][src]),
                 "conf": rel_l2(o["conf"][slot].cpu().numpy()[::sub, ::sub], g[f"{side}_conf"][src]),
                 "pose": rel_l2(o["relative_pose"][slot].cpu().numpy(), g[f"{side}_pose"][src]),
                 "pose_conf": rel_l2(o["relative_pose_conf"][slot].cpu().numpy(), g[f"{side}_pose_conf"][src])}
            bad = {k: v for k, v in e.items() if v > TOL}
            assert not bad, (slot, src, side, bad)


def np_concat(x, y):
    import numpy as np
    return np.ascontiguousarray(np.concatenate([x, y], 0))


@pytest.mark.parametrize("prec", [DEFAULT, "f16x3"])
def test_u8_hwc_input_matches_normalised_fp32(G, prec):
    """f3 (input step): uint8 HWC frames with the reference ImgNorm fused into the patch gather give
    exactly the encoder features / outputs of the fp32 NCHW path."""
    import torch
    from vista_slam_amd import weights as W
    m = G.model("tiny", 1.0, prec)
    G.set_variant(m, 0)
    f32 = torch.from_numpy(W.synth_images(4, 48, 64, seed=43, tag=5)).cuda()
    u8 = torch.from_numpy(W.synth_images_u8(4, 48, 64, seed=43, tag=5)).cuda()
    fa, _ = m._encode_image(f32, None, normalize=False)
    fb, _ = m.encode_u8hwc(u8)
    torch.cuda.synchronize()
    # the gathered patches are bit-identical and the default mode is bit-reproducible (slab split-K, no atomics, since round 2);
    # the bound stays a tolerance so that the test does not pin the two entry points to the same tile path
    from helpers import rel_l2
    assert rel_l2(fb.cpu().numpy(), fa.cpu().numpy()) < 2e-6
    m1, s1 = m.forward_pair(f32[:2], f32[2:])
    m2, s2 = m.forward_pair_u8hwc(u8[:2], u8[2:])
    torch.cuda.synchronize()
    for k in ("pts3d_pred", "conf", "relative_pose", "relative_pose_conf"):
        assert rel_l2(m2[k].cpu().numpy(), m1[k].cpu().numpy()) < 1e-5 and rel_l2(s2[k].cpu().numpy(), s1[k].cpu().numpy()) < 1e-5


@pytest.mark.parametrize("prec", [DEFAULT, "f16x3"])
def test_calls_on_two_streams_overlap_safely(G, prec):
    """One scratch context per caller stream (sta_mi355.h "Streams and concurrency"): `_encode_image` of the NEXT keyframe
    enqueued on a second stream while `regress_views` of the current one runs on the first (the SLAM loop's own independence,
    slam.py:258 vs :263-277) must give, bit for bit, what the two calls give one after the other - repeatedly, with the
    roles of the streams swapped, and with forward_pair on the second stream as well."""
    import torch
    from vista_slam_amd import weights as W
    from vista_slam_amd.slam_scheduler import regress_views
    m = G.model("full", 1.0, prec)
    G.set_variant(m, 0)
    H = Wd = 224
    imgs = torch.from_numpy(W.synth_images(6, H, Wd, seed=43, tag=17)).cuda()
    feats = [m._encode_image(imgs[v:v + 1], None, normalize=False)[0] for v in range(4)]
    torch.cuda.synchronize()
    # serial references
    ref_feat = m._encode_image(imgs[4:5], None, normalize=False)[0].clone()
    ref_edges = regress_views(m, feats[3], feats[:3], [False, False, True], -1.0, H, Wd)
    ref_pair = m.forward_pair(imgs[4:5], imgs[5:6])
    torch.cuda.synchronize()
    ref_d = [r.depths.clone() for r in ref_edges]; ref_p = [r.pose.clone() for r in ref_edges]
    ref_pts = ref_pair[0]["pts3d_pred"].clone()
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    for it in range(4):
        sa, sb = (s1, s2) if it % 2 == 0 else (s2, s1)
        with torch.cuda.stream(sb):          # enqueued first: runs under the scheduler call below
            f = m._encode_image(imgs[4:5], None, normalize=False)[0]
            pr = m.forward_pair(imgs[4:5], imgs[5:6]) if it >= 2 else None
        with torch.cuda.stream(sa):
            edges = regress_views(m, feats[3], feats[:3], [False, False, True], -1.0, H, Wd)
        torch.cuda.synchronize()
        assert torch.equal(f, ref_feat), f"encode on a second stream differs (iteration {it})"
        for r, d, p_ in zip(edges, ref_d, ref_p):
            assert torch.equal(r.depths, d) and torch.equal(r.pose, p_), f"scheduler under a concurrent encode differs (iteration {it})"
        if pr is not None:
            assert torch.equal(pr[0]["pts3d_pred"], ref_pts), f"forward_pair on a second stream differs (iteration {it})"
    assert m.range_report() == (0, 0)


@pytest.mark.parametrize("prec", [DEFAULT, "f16x3"])
def test_dpt_side_lane_is_bit_identical(G, prec):
    """dpt_impl's side lane (the branch kernels of the DPT head on an internal second stream, SLAM-scale calls only) runs the
    same kernels with the same split-K slices as the one-lane order: outputs bit for bit equal with the lanes switched off
    (sta_set_side_lanes: the application's switch), for forward_pair and for a scheduler call, repeatedly (races would show as
    flakiness)."""
    import torch
    from vista_slam_amd import weights as W, _lib
    from vista_slam_amd.slam_scheduler import regress_views
    m = G.model("full", 1.0, prec)
    G.set_variant(m, 0)
    H = Wd = 224
    imgs = torch.from_numpy(W.synth_images(6, H, Wd, seed=43, tag=23)).cuda()
    feats = [m._encode_image(imgs[v:v + 1], None, normalize=False)[0] for v in range(4)]
    try:
        m.set_side_lanes("off")
        ref = m.forward_pair(imgs[4:5], imgs[5:6])
        ref_pts, ref_conf = ref[0]["pts3d_pred"].clone(), ref[1]["conf"].clone()
        ref_e = regress_views(m, feats[3], feats[:3], [False, False, True], -1.0, H, Wd)
        ref_d = [r.depths.clone() for r in ref_e]
        torch.cuda.synchronize()
        m.set_side_lanes("on")
        for it in range(5):
            out = m.forward_pair(imgs[4:5], imgs[5:6])
            e = regress_views(m, feats[3], feats[:3], [False, False, True], -1.0, H, Wd)
            torch.cuda.synchronize()
            assert torch.equal(out[0]["pts3d_pred"], ref_pts) and torch.equal(out[1]["conf"], ref_conf), f"forward_pair differs with the side lane (iteration {it})"
            for r, d in zip(e, ref_d):
                assert torch.equal(r.depths, d), f"scheduler differs with the side lane (iteration {it})"
    finally:
        m.set_side_lanes("auto")
    assert m.range_report() == (0, 0)


def test_stream_contexts_are_recycled(G):
    """A handle keeps at most 8 scratch contexts; a call on one more stream takes over the least recently used context
    (sta_mi355.h "Streams and concurrency") instead of failing - a long-lived process that keeps creating streams stays usable."""
    import torch
    from vista_slam_amd import weights as W
    m = G.model("tiny", 1.0, DEFAULT)
    G.set_variant(m, 0)
    imgs = torch.from_numpy(W.synth_images(2, 64, 64, seed=43, tag=3)).cuda()
    ref = m.forward_pair(imgs[:1], imgs[1:])[0]["pts3d_pred"].clone()
    streams = [torch.cuda.Stream() for _ in range(12)]
    assert len({s.cuda_stream for s in streams}) > 8, "the test needs more than 8 distinct streams"
    for s in streams:
        with torch.cuda.stream(s):
            out = m.forward_pair(imgs[:1], imgs[1:])[0]["pts3d_pred"]
        s.synchronize()
        assert torch.equal(out, ref)


def test_split_phase_scheduler_pipelines_two_keyframes(G):
    """sta_regress_views_begin / _finish: decode + pose heads of keyframe B's edges enqueued (second stream) BEFORE keyframe A's
    accept / reject + DPT heads run (first stream) - the pipelined schedule of bench.slam_replay - gives bit for bit what the
    two plain regress_views calls give; a second call on a stream with a pending begin fails loudly, and so does a finish
    without a begin."""
    import torch
    from vista_slam_amd import _lib, weights as W
    from vista_slam_amd.slam_scheduler import regress_views, regress_views_begin, regress_views_finish
    m = G.model("full", 1.0, DEFAULT)
    G.set_variant(m, 0)
    H = Wd = 224
    imgs = torch.from_numpy(W.synth_images(6, H, Wd, seed=43, tag=19)).cuda()
    feats = [m._encode_image(imgs[v:v + 1], None, normalize=False)[0] for v in range(6)]
    ref_a = regress_views(m, feats[4], feats[1:4], [False, False, True], -1.0, H, Wd)
    confs = sorted(r.rel_pose_conf for r in ref_a[:2])
    thres = 0.5 * (confs[0] + confs[1])                       # rejects one of the two non-adjacent edges
    ref_a = regress_views(m, feats[4], feats[1:4], [False, False, True], thres, H, Wd)
    ref_b = regress_views(m, feats[5], feats[2:5], [False, False, True], -1.0, H, Wd)
    torch.cuda.synchronize()
    assert [r.accepted for r in ref_a].count(False) == 1
    keep = [(r.accepted, r.pose.clone(), None if not r.accepted else (r.depths.clone(), r.confs.clone(), r.intri.clone())) for r in ref_a + ref_b]
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    for _ in range(2):
        with torch.cuda.stream(s1):
            pa = regress_views_begin(m, feats[4], feats[1:4], H, Wd)
        with torch.cuda.stream(s2):
            pb = regress_views_begin(m, feats[5], feats[2:5], H, Wd)
        with torch.cuda.stream(s1):
            with pytest.raises(_lib.StaError, match="pending"):
                m._encode_image(imgs[:1], None, normalize=False)           # this stream's scratch context is live
            with pytest.raises(_lib.StaError, match="not been finished"):
                regress_views_begin(m, feats[5], feats[2:5], H, Wd)
            got_a = regress_views_finish(m, pa, [False, False, True], thres)
        with torch.cuda.stream(s2):
            got_b = regress_views_finish(m, pb, [False, False, True], -1.0)
        torch.cuda.synchronize()
        for r, (acc, pose, rest) in zip(got_a + got_b, keep):
            assert r.accepted == acc and torch.equal(r.pose, pose)
            if acc:
                assert torch.equal(r.depths, rest[0]) and torch.equal(r.confs, rest[1]) and torch.equal(r.intri, rest[2])
    with torch.cuda.stream(s1):
        with pytest.raises(AssertionError, match="already finished"):
            regress_views_finish(m, pa, [False, False, True], thres)
        m._encode_image(imgs[:1], None, normalize=False)                   # and the stream is usable again
    torch.cuda.synchronize()


def test_abandoned_scheduler_call_releases_its_stream(G):
    """a begin that is never finished (an exception between the phases, a dropped PendingEdges) must not leave its
    stream unusable.  PendingEdges.close() / `with` / garbage collection call sta_regress_views_abort; nine abandoned calls on
    nine streams no longer exhaust the handle's eight scratch contexts; an aborted call cannot be finished."""
    import gc
    import torch
    from vista_slam_amd import _lib, weights as W
    from vista_slam_amd.slam_scheduler import regress_views, regress_views_begin, regress_views_finish
    m = G.model("tiny", 1.0, DEFAULT)
    H, Wd = 48, 64
    imgs = torch.from_numpy(W.synth_images(3, H, Wd, seed=43, tag=29)).cuda()
    feats = [m._encode_image(imgs[v:v + 1], None, normalize=False)[0] for v in range(3)]
    ref = regress_views(m, feats[2], feats[:2], [False, True], -1.0, H, Wd)
    torch.cuda.synchronize()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        p = regress_views_begin(m, feats[2], feats[:2], H, Wd)
        with pytest.raises(_lib.StaError, match="pending"):
            m._encode_image(imgs[:1], None, normalize=False)
        p.close()                                                          # explicit abort
        p.close()                                                          # idempotent
        m._encode_image(imgs[:1], None, normalize=False)                   # the stream is usable again
        with pytest.raises(AssertionError, match="already finished or aborted"):
            regress_views_finish(m, p, [False, True], -1.0)
        try:
            with regress_views_begin(m, feats[2], feats[:2], H, Wd):
                raise KeyError("host-side failure between the phases")
        except KeyError:
            pass
        got = regress_views(m, feats[2], feats[:2], [False, True], -1.0, H, Wd)      # context manager released it
    torch.cuda.synchronize()
    for a, b in zip(got, ref):
        assert torch.equal(a.pose, b.pose) and torch.equal(a.depths, b.depths)
    streams = [torch.cuda.Stream() for _ in range(9)]
    for st_ in streams:                                                    # nine dropped begins: each released by __del__
        with torch.cuda.stream(st_):
            q = regress_views_begin(m, feats[2], feats[:2], H, Wd)
            del q
            gc.collect()
    with torch.cuda.stream(streams[0]):
        got = regress_views(m, feats[2], feats[:2], [False, True], -1.0, H, Wd)
    torch.cuda.synchronize()
    assert torch.equal(got[1].depths, ref[1].depths)


def _pre_goldens():
    import glob
    import os
    return sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "pre_*.npz")))


@pytest.mark.parametrize("path", _pre_goldens(), ids=[p.split("pre_")[-1][:-4] for p in _pre_goldens()])
def test_input_step_f3_bit_exact_to_pillow_goldens(G, path):
    """f3: fused crop -> LANCZOS -> crop -> ImgNorm / ImgGray kernels == real Pillow + reference cropping goldens,
    bit for bit (uint8 image and both float tensors)."""
    import numpy as np
    import torch
    from vista_slam_amd import weights as W
    from vista_slam_amd.preprocess import process_image
    g = np.load(path)
    Hs, Ws = (int(v) for v in g["src_hw"])
    ow, oh = (int(v) for v in g["target_wh"])
    src = W.synth_frames_u8(Hs, Ws, seed=43, tag=int(g["tag"]))
    m = G.model("tiny", 1.0, "f16x3")
    out = process_image(m, src, (ow, oh), img_name="/some/dir/frame.png")
    torch.cuda.synchronize()
    assert out["img_name"] == "frame.png"
    assert np.array_equal(out["u8"].cpu().numpy(), g["u8"])
    assert np.array_equal(out["rgb"].cpu().numpy()[:, ::5, ::5], g["rgb_s"])
    assert np.array_equal(out["gray"].cpu().numpy()[:, ::5, ::5], g["gray_s"])
    assert abs(float(out["rgb"].double().sum()) - float(g["rgb_sum"])) < 1e-6 * out["rgb"].numel()


@pytest.mark.parametrize("prec", [DEFAULT, "f16x3"])
def test_input_step_f3_vs_oracle_random_geometries_and_encoder_handoff(G, prec):
    """Random source sizes / edges / targets against the oracle (bit-exact), the cached-table switch between
    geometries, error cases, and the hand-off: encode_u8hwc(u8) == _encode_image(rgb)."""
    import numpy as np
    import torch
    from helpers import rel_l2
    from oracle import preprocess_oracle as P
    from vista_slam_amd import weights as W
    from vista_slam_amd.preprocess import process_image
    m = G.model("tiny", 1.0, prec)
    rng = np.random.default_rng(7)
    for it in range(8):
        Hs = int(rng.integers(120, 700)); Ws = int(Hs * rng.uniform(1.15, 2.2))
        ow = 16 * int(rng.integers(4, 20)); oh = 16 * int(rng.integers(3, 1 + ow // 16))
        if 0.9 < (Hs - 8) / (Ws - 8) < 1.1 and ow != oh:
            continue
        we, he = int(rng.integers(0, 12)), int(rng.integers(0, 12))
        src = W.synth_frames_u8(Hs, Ws, seed=43, tag=50 + it)
        want = P.process_image(src, ow, oh, we, he)
        for _ in range(2):                                   # second call takes the cached-table path
            out = process_image(m, src, (ow, oh), we, he)
        torch.cuda.synchronize()
        assert np.array_equal(out["u8"].cpu().numpy(), want["u8"]), (Hs, Ws, ow, oh, we, he)
        assert np.array_equal(out["rgb"].cpu().numpy(), want["rgb"])
        assert np.array_equal(out["gray"].cpu().numpy(), want["gray"])
    src = W.synth_frames_u8(120, 160, seed=43, tag=70)
    out = process_image(m, src, (64, 48))
    fa, _ = m.encode_u8hwc(out["u8"][None])
    fb, _ = m._encode_image(out["rgb"][None], None, normalize=False)
    torch.cuda.synchronize()
    assert rel_l2(fa.cpu().numpy(), fb.cpu().numpy()) < 2e-6
    for Hs, Ws, res in ((300, 200, (64, 48)), (333, 290, (80, 80)), (512, 200, (96, 32))):     # portrait frames: transposed resolution
        srcp = W.synth_frames_u8(Hs, Ws, seed=43, tag=71 + Hs)
        want = P.process_image(srcp, res[0], res[1])
        out = process_image(m, srcp, res)
        torch.cuda.synchronize()
        assert tuple(out["u8"].shape) == (res[0], res[1], 3) and tuple(out["rgb"].shape) == (3, res[0], res[1])
        assert np.array_equal(out["u8"].cpu().numpy(), want["u8"]) and np.array_equal(out["rgb"].cpu().numpy(), want["rgb"])
        assert np.array_equal(out["gray"].cpu().numpy(), want["gray"])
    with pytest.raises(RuntimeError, match="square frame"):
        process_image(m, W.synth_frames_u8(200, 200, seed=43, tag=72), (64, 48))
    with pytest.raises(RuntimeError, match="landscape or square"):
        process_image(m, src, (48, 64))


@pytest.mark.parametrize("prec", [DEFAULT, "f16x3"])
def test_output_step_f4_pointcloud_and_files(G, tmp_path, prec):
    """f4: world point cloud vs the reference golden (compute_local_pointclouds + bmm + mask order), PLY records,
    the save_data_all file set re-read the way eval_recon.load_data does, and mat -> SE3."""
    import numpy as np
    import torch
    from helpers import load_golden, max_rel
    from vista_slam_amd import formats as F
    m = G.model("tiny", 1.0, prec)
    g = load_golden("fmt")[0]
    pts, col, rec = F.world_pointcloud(m, g["depths"], g["scales"], g["intrinsics"], g["poses"], g["confs"], g["imgs"],
                                       float(g["thres"]), want_records=True)
    assert pts.shape == g["points"].shape and len(rec) == len(g["points"])
    assert max_rel(pts.cpu().numpy(), g["points"]) < 1e-5
    assert np.abs(col.cpu().numpy() - g["colors"]).max() < 1e-6
    assert np.array_equal(rec["x"], pts[:, 0].cpu().numpy().astype(np.float64))
    assert np.array_equal(rec["blue"], np.rint(np.clip(col[:, 2].cpu().numpy(), 0, 1) * 255).astype(np.uint8))
    out = str(tmp_path / "run")
    vg = {0: [1], 1: [0, 2], 2: [1]}
    F.save_data_all(m, out, poses=g["poses"], scales=g["scales"], depths=g["depths"], confs=g["confs"],
                    intrinsics=g["intrinsics"], imgs=g["imgs"], conf_thres=float(g["thres"]), view_graph=vg,
                    loop_min_dist=40, view_names=["a.png", "b.png", "c.png"], gt_poses=[np.eye(4)] * 3)
    z = np.load(out + "/view_graph.npz", allow_pickle=True)                     # eval_recon.py:13-16
    assert z["view_graph"].item() == vg and z["loop_min_dist"].item() == 40 and z["view_names"].tolist() == ["a.png", "b.png", "c.png"]
    assert np.array_equal(np.load(out + "/trajectory.npy"), g["poses"]) and np.load(out + "/scales.npy")[..., None].shape == (3, 1, 1)
    assert np.array_equal(np.load(out + "/depths.npy"), g["depths"]) and np.load(out + "/intrinsics.npy").shape == (3, 3, 3)
    c = np.load(out + "/confs.npz")
    assert np.array_equal(c["confs"], g["confs"]) and abs(c["thres"].item() - float(g["thres"])) < 1e-12
    im = np.load(out + "/images.npy")
    assert im.shape == (3, 24, 32, 3) and 0.0 <= im.min() and im.max() <= 1.0
    assert np.load(out + "/gt_poses.npy").dtype == np.float32
    back = F.read_ply(out + "/pointcloud.ply")
    assert np.array_equal(back, rec)
    # mat -> SE3: translation copied, unit quaternion reproducing R, qw >= 0
    se3 = F.mat_to_se3(m, g["poses"]).cpu().numpy()
    assert np.array_equal(se3[:, :3], g["poses"][:, :3, 3])
    x, y, zq, w = se3[:, 3], se3[:, 4], se3[:, 5], se3[:, 6]
    R = np.stack([1 - 2 * (y * y + zq * zq), 2 * (x * y - zq * w), 2 * (x * zq + y * w),
                  2 * (x * y + zq * w), 1 - 2 * (x * x + zq * zq), 2 * (y * zq - x * w),
                  2 * (x * zq - y * w), 2 * (y * zq + x * w), 1 - 2 * (x * x + y * y)], 1).reshape(-1, 3, 3)
    assert np.abs(R - g["poses"][:, :3, :3]).max() < 2e-6 and (w >= 0).all()
    q = g["quat_xyzw"] * np.sign(g["quat_xyzw"][:, 3:4])
    assert np.abs(se3[:, 3:] - q).max() < 2e-6


def _sequential_regress(m, feats, pos, i, j, thres, H, Wd):
    """The reference call pattern of regress_two_views (slam.py:153-189) through the drop-in shim, one edge, B=1."""
    import torch
    from vista_slam_amd import post
    d_ij, d_ji = m._decode_stereo(feats[i], feats[j], pos, pos)
    pose = m.head_pose_s(d_ij[-1][:, 0, :])
    c = float(pose["conf"][0])
    if c < thres and i - j != 1:
        return pose["pose"][0], c, None, None, None
    ts = torch.tensor([[H, Wd]])
    ji = m.head_pts([feats[j]] + [t[:, 1:, :] for t in d_ji], ts)
    ij = m.head_pts([feats[i]] + [t[:, 1:, :] for t in d_ij], ts)
    pcls = torch.cat([ij["pts3d"], ji["pts3d"]], 0)
    confs = torch.cat([ij["conf"], ji["conf"]], 0)
    intri = post.estimate_intrinsic_from_pts3d(m, pcls, confs, shared_intrinsic=True)
    return pose["pose"][0], c, confs, intri, pcls[..., 2]


@pytest.mark.parametrize("cfg,H,Wd,nview", [("tiny", 48, 64, 6), ("full", 224, 224, 4)])
@pytest.mark.parametrize("prec", [DEFAULT, "f16x3"])
def test_keyframe_scheduler_f2_matches_sequential_edges(G, cfg, H, Wd, nview, prec):
    """f2: one batched sta_regress_views call == the reference's per-edge regress_two_views sequence, incl. the
    early reject (threshold set between the observed confidences so both branches are exercised) and the
    adjacent-edge exemption."""
    import torch
    from helpers import rel_l2
    from vista_slam_amd import weights as W
    from vista_slam_amd.slam_scheduler import regress_views
    m = G.model(cfg, 1.0, prec)
    G.set_variant(m, 0)
    imgs = torch.from_numpy(W.synth_images(nview, H, Wd, seed=43, tag=21)).cuda()
    feats, pos = [], None
    for v in range(nview):
        f, pos = m._encode_image(imgs[v:v + 1], None, normalize=False)
        feats.append(f)
    i = nview - 1
    js = list(range(i))
    probe = [_sequential_regress(m, feats, pos, i, j, -1.0, H, Wd)[1] for j in js]
    thres = float(sorted(probe)[len(probe) // 2]) + 1e-7          # rejects about half of the non-adjacent edges
    seq = [_sequential_regress(m, feats, pos, i, j, thres, H, Wd) for j in js]
    res = regress_views(m, feats[i], [feats[j] for j in js], [i - j == 1 for j in js], thres, H, Wd)
    torch.cuda.synchronize()
    n_rej = 0
    for j, r, (pose, c, confs, intri, depths) in zip(js, res, seq):
        assert abs(r.rel_pose_conf - c) <= 2e-6 * max(1.0, abs(c)), (j, r.rel_pose_conf, c)
        assert rel_l2(r.pose.cpu().numpy(), pose.cpu().numpy()) < 2e-5
        assert r.accepted == (confs is not None), (j, r.rel_pose_conf, thres)
        if confs is None:
            n_rej += 1
            assert r.confs is None and r.intri is None and r.depths is None
            continue
        # (the batched call picks other tile families / split-K forms than the B = 1 calls: same arithmetic, other summation
        # orders; in the default policy the DPT head's fp8 correction term makes that a 1e-5-class difference)
        assert rel_l2(r.confs.cpu().numpy(), confs.cpu().numpy()) < ctol(prec)
        assert rel_l2(r.depths.cpu().numpy(), depths.cpu().numpy()) < ctol(prec)
        assert rel_l2(r.intri.cpu().numpy(), intri.cpu().numpy()) < ctol(prec)
    assert res[-1].accepted                      # the adjacent edge is never rejected (slam.py:169)
    assert 0 < n_rej < len(js)


@pytest.mark.parametrize("name,cfg", [("f2_tiny_48x64", "tiny"), ("f2_tiny_80x48_portrait", "tiny"), ("f2_full_224", "full")])
@pytest.mark.parametrize("prec", [DEFAULT, "f16x3"])
def test_keyframe_scheduler_f2_vs_reference_golden(G, name, cfg, prec):
    """f2 pinned by the reference: sta_regress_views (one batched call for all edges of the keyframe) and the per-edge
    split calls vs tests/golden/f2_*.npz = regress_two_views (slam.py:153-189) replayed on the reference model with the
    reference's estimate_intrinsic_from_pts3d (oracle/gen_golden.py gen_f2): accepted / rejected edges and the
    adjacent-edge exemption of slam.py:169.  The portrait case: maps come back as the transposed views [2,W,H] the
    reference works on, and K is what its estimate_intrinsic_from_pts3d derives from those views."""
    import numpy as np
    import torch
    from helpers import load_golden, rel_l2, max_rel
    from vista_slam_amd import weights as W
    from vista_slam_amd.slam_scheduler import regress_views
    if cfg == "full":
        G.drop_models()
    g, meta = load_golden(name)
    H, Wd, nview, sub = int(meta["H"]), int(meta["W"]), int(meta["nview"]), int(meta["sub"])
    m = G.model(cfg, 1.0, prec)
    G.set_variant(m, 0)
    imgs = torch.from_numpy(W.synth_images(nview, H, Wd, seed=int(meta["seed"]), tag=int(meta["tag"]))).cuda()
    feats = [m._encode_image(imgs[v:v + 1], None, normalize=False)[0] for v in range(nview)]
    i = nview - 1
    js = list(range(i))
    thres = float(g["thres"])
    res = regress_views(m, feats[i], [feats[j] for j in js], [i - j == 1 for j in js], thres, H, Wd)
    torch.cuda.synchronize()
    acc = g["accepted"]
    assert acc.any() and not acc.all()
    for j, r in zip(js, res):
        assert abs(r.rel_pose_conf - float(g[f"conf_{j}"])) < 1e-4, (j, r.rel_pose_conf, float(g[f"conf_{j}"]))
        assert rel_l2(r.pose.cpu().numpy(), g[f"pose_{j}"]) < TOL
        assert r.accepted == bool(acc[j]), (j, r.rel_pose_conf, thres)
        if not r.accepted:
            assert r.confs is None and r.intri is None and r.depths is None
            continue
        confs, depths = r.confs.cpu().numpy(), r.depths.cpu().numpy()
        assert rel_l2(confs[:, ::sub, ::sub], g[f"confs_{j}"]) < TOL
        assert rel_l2(depths[:, ::sub, ::sub], g[f"depths_{j}"]) < TOL
        assert abs(np.sqrt((confs.astype(np.float64) ** 2).sum()) / float(g[f"confs_l2_{j}"]) - 1) < TOL
        assert abs(np.sqrt((depths.astype(np.float64) ** 2).sum()) / float(g[f"depths_l2_{j}"]) - 1) < TOL
        assert max_rel(r.intri.cpu().numpy(), g[f"intri_{j}"]) < TOL
    assert res[-1].accepted and float(g[f"conf_{i - 1}"]) < thres      # accepted only through the adjacency exemption
    # _encode_image(normalize=True): the reference's default argument (sta_model.py:163,172-173)
    fn, _ = m._encode_image(imgs[i:i + 1], None, normalize=True)
    assert rel_l2(fn.cpu().numpy()[:, ::max(1, sub)], g["enc_feat_norm"]) < TOL
    if cfg == "full":
        G.drop_models()


@pytest.mark.parametrize("prec", [DEFAULT, "f16x3"])
def test_forward_encodes_main_view_once_and_matches_forward_pair(G, prec):
    """forward(views) (sta_model.py:247-291) with two support views == forward_pair per support view (the main view is
    encoded once, sta_model.py:257)."""
    import torch
    from helpers import rel_l2
    from vista_slam_amd import weights as W
    m = G.model("tiny", 1.0, prec)
    imgs = torch.from_numpy(W.synth_images(3, 48, 64, seed=43, tag=31)).cuda()
    views = {"main_view": {"img": imgs[0:1]}, "neighbor_views": [{"img": imgs[1:2]}], "loop_views": [{"img": imgs[2:3]}]}
    calls = []
    orig = m._encode_image
    m._encode_image = lambda *a, **k: (calls.append(1), orig(*a, **k))[1]
    try:
        out = m(views)
    finally:
        m._encode_image = orig
    assert len(calls) == 2                                   # the main view once + ONE batched call for all support views
    for k, v in enumerate((imgs[1:2], imgs[2:3])):
        mm, ss = m.forward_pair(imgs[0:1], v)
        for key in ("pts3d_pred", "conf", "relative_pose", "relative_pose_conf"):
            assert rel_l2(out["main_views"][k][key].cpu().numpy(), mm[key].cpu().numpy()) < 2e-5, (k, key)
            assert rel_l2(out["support_views"][k][key].cpu().numpy(), ss[key].cpu().numpy()) < 2e-5, (k, key)


def test_keyframe_scheduler_f2_all_rejected_and_errors(G):
    import torch
    from vista_slam_amd import weights as W
    from vista_slam_amd.slam_scheduler import regress_views
    m = G.model("tiny", 1.0, "f16x3")
    imgs = torch.from_numpy(W.synth_images(3, 48, 64, seed=43, tag=22)).cuda()
    feats = [m._encode_image(imgs[v:v + 1], None, normalize=False)[0] for v in range(3)]
    res = regress_views(m, feats[2], [feats[0]], [False], 2.0, 48, 64)       # sigmoid conf < 2: always rejected
    assert not res[0].accepted and res[0].depths is None and 0.0 < res[0].rel_pose_conf < 1.0
    with pytest.raises(AssertionError):
        regress_views(m, feats[2], [feats[0][:, :5]], [False], 0.5, 48, 64)


@pytest.mark.parametrize("prec", [DEFAULT, "f16x3"])
def test_post_sta_reductions_f1(G, prec):
    """f1: fused intrinsics / depth / mean-confidence pass and the scale estimate vs reference goldens."""
    import torch
    from helpers import load_golden, max_rel
    from vista_slam_amd import post
    m = G.model("tiny", 1.0, prec)
    g = load_golden("post")[0]
    pts, conf = torch.from_numpy(g["pts"]).cuda(), torch.from_numpy(g["conf"]).cuda()
    K, depth, cmean = post.pair_reductions(m, pts, conf, shared_intrinsic=True)
    Kp = post.estimate_intrinsic_from_pts3d(m, pts, conf, shared_intrinsic=False)
    s = post.estimate_scale_with_depth_and_confidence(m, pts[0, ..., 2], pts[1, ..., 2], conf[0], conf[1])
    torch.cuda.synchronize()
    assert max_rel(K.cpu().numpy(), g["K_shared"]) < 1e-5
    assert max_rel(Kp.cpu().numpy(), g["K_per"]) < 1e-5
    assert torch.equal(depth, pts[..., 2])
    assert max_rel(cmean.cpu().numpy(), g["conf_mean"]) < 1e-6
    assert abs(float(s) - float(g["scale"])) < 1e-5 * abs(float(g["scale"]))


@pytest.mark.parametrize("prec", [DEFAULT, "f16x3"])
def test_multi_gpu_layer_on_real_outputs_nccl_world1(G, prec):
    """SURVEY 8(e) on the GPU: pack_compact -> gather_compact (RCCL all_gather_into_tensor) -> unpack_compact on REAL
    forward_pair outputs and pack_edges -> gather_edges on REAL regress_views results, in an `nccl` process group of
    world size 1 (one GPU per box; the world-2 paths run under gloo in tests/test_dist_cpu.py)."""
    import os
    import socket
    import torch
    import torch.distributed as dist
    from vista_slam_amd import parallel as P
    from vista_slam_amd import weights as W
    from vista_slam_amd.slam_scheduler import regress_views
    m = G.model("tiny", 1.0, prec)
    H, Wd, B = 48, 64, 3
    imgs = torch.from_numpy(W.synth_images(2 * B, H, Wd, seed=43, tag=41)).cuda()
    main_o, supp_o = m.forward_pair(imgs[:B], imgs[B:])
    feats = [m._encode_image(imgs[v:v + 1], None, normalize=False)[0] for v in range(4)]
    edges = regress_views(m, feats[3], [feats[j] for j in range(3)], [False, False, True], 0.0, H, Wd)
    edges_rej = regress_views(m, feats[3], [feats[0]], [False], 2.0, H, Wd)     # sigmoid conf < 2: rejected
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1,
                            device_id=torch.device("cuda:0"))
    try:
        assert dist.get_backend() == "nccl"
        local = P.pack_compact(main_o, supp_o)
        assert local.shape == (B, P.compact_elems_per_pair(H, Wd))
        allp = P.gather_compact(local, B)
        torch.cuda.synchronize()
        assert allp.data_ptr() != local.data_ptr()                       # went through the collective
        um, us = P.unpack_compact(allp, H, Wd)
        for u, o in ((um, main_o), (us, supp_o)):
            assert torch.equal(u["relative_pose"], o["relative_pose"]) and torch.equal(u["relative_pose_conf"], o["relative_pose_conf"])
            assert torch.equal(u["depth"], o["pts3d_pred"][..., 2]) and torch.equal(u["conf"], o["conf"])
        res = P.gather_edges(P.pack_edges(edges + edges_rej, H, Wd), 4, H, Wd)
        torch.cuda.synchronize()
        for r, e in zip(res, edges + edges_rej):
            assert r["accepted"] == e.accepted and abs(r["rel_pose_conf"] - e.rel_pose_conf) < 1e-7
            assert torch.equal(r["pose"], e.pose)
            if e.accepted:
                assert torch.equal(r["depths"], e.depths) and torch.equal(r["confs"], e.confs) and torch.equal(r["intri"], e.intri)
            else:
                assert r["depths"] is None and r["confs"] is None and r["intri"] is None
        assert not res[3]["accepted"] and all(r["accepted"] for r in res[:3])
    finally:
        dist.destroy_process_group()


def test_deterministic_mode_is_bit_reproducible_at_slam_scale(G):
    """The B = 1 @224x224 split entry points (the regime whose GEMMs split K over workgroups) give identical bits on repeated
    runs - in the default mode (K slices are combined in a fixed order: slabs) and with sta_set_deterministic, which takes the
    same path on the product shapes."""
    import torch
    from helpers import rel_l2
    from vista_slam_amd import weights as W
    G.drop_models()
    m = G.model("full", 1.0, "f16x3h")
    imgs = torch.from_numpy(W.synth_images(2, 224, 224, seed=43, tag=51)).cuda()
    ts = torch.tensor([[224, 224]])

    def run():
        fa, pa = m._encode_image(imgs[:1], ts, normalize=False)
        fb, pb = m._encode_image(imgs[1:], ts, normalize=False)
        d1, d2 = m._decode_stereo(fa, fb, pa, pb, layers=[6, 9, 12])
        pts = m.head_pts([fa] + [None if t is None else t[:, 1:, :] for t in d1], ts)
        pose = m.head_pose_s(d1[-1][:, 0, :])
        torch.cuda.synchronize()
        return fa.clone(), d1[-1].clone(), pts["pts3d"].clone(), pose["conf"].clone()
    ref, ref2 = run(), run()
    for x, y in zip(ref, ref2):
        assert torch.equal(x, y)                # default mode: no atomics on the product path any more
    m.set_deterministic(True)
    try:
        a, b = run(), run()
        for x, y in zip(a, b):
            assert torch.equal(x, y)
        for x, y in zip(a, ref):
            assert torch.equal(x, y) or rel_l2(x.cpu().numpy(), y.cpu().numpy()) < 1e-4
    finally:
        m.set_deterministic(False)
    G.drop_models()


def test_decode_stereo_unequal_token_counts_is_refused(G):
    """`_decode_stereo` with N1 != N2 (two views of different resolution): allowed by the reference's module code
    (sta_blocks.py:193-205), never produced by slam.py / forward (sta_model.py:257-262); the batched decoder here needs equal
    token grids and says so instead of padding silently."""
    import torch
    m = G.model("tiny", 1.0, DEFAULT)
    E = m.cfg.enc_embed_dim
    f1 = torch.zeros(1, 12, E, device="cuda"); f2 = torch.zeros(1, 8, E, device="cuda")
    p1 = torch.zeros(1, 12, 2, dtype=torch.int64, device="cuda"); p2 = torch.zeros(1, 8, 2, dtype=torch.int64, device="cuda")
    with pytest.raises(AssertionError, match="same token grid"):
        m._decode_stereo(f1, f2, p1, p2)


# ---------------------------------------------------------------------------------------------------------
# BASELINE configs[2-3] as a workload: the frontend call sequence of OnlineSLAM.step over a growing feature cache, in the
# tumrgbd.yaml (3 neighbour + <= 2 loop edges), 7scenes.yaml (2 + 3) and default.yaml (3 + 3: the ScanNet runs) edge regimes, against REFERENCE goldens
# (oracle/gen_golden.py gen_seq: add_view + connect_view_i_j replayed on the imported reference model, slam.py:142-241,244-297).
SEQ_TINY = ["seq_tum_tiny_48x64", "seq_7scenes_tiny_48x64", "seq_tum_tiny_48x64_t075", "seq_default_tiny_48x64"]
SEQ_FULL = ["seq_tum_full_224", "seq_7scenes_full_224", "seq_default_full_224",
            "seq_tum_full_224_t075", "seq_tum_full_224_sharp"]      # round 6: the yamls' own rel_pose_thres at full size; Q/K gain 3


def _seq_run(G, case, schedule, prec=DEFAULT, frontend=None):
    import numpy as np
    import torch
    from helpers import load_golden, seq_meta
    from oracle.seq_protocol import seq_edge_list, seq_frames
    from vista_slam_amd import weights as W
    from vista_slam_amd.keyframe_pipeline import replay
    g, meta = load_golden(case)
    sm = seq_meta(meta)
    H, Wd = sm["H"], sm["W"]
    if frontend is not None:           # the real-checkpoint kit: weights from a file
        m = frontend
        m.set_precision(prec)
    else:
        m = G.model("tiny" if "tiny" in case else "full", float(meta.get("qk_gain", 1.0)), prec, sm["seed"])
    m.range_report(reset=True)
    frames = torch.from_numpy(seq_frames(W, sm["nkf"], H, Wd, sm["seed"], sm["tag"])).cuda()
    ts = torch.tensor([[H, Wd]])

    def add_view(i):                                               # slam.py:142-151
        return m._encode_image(frames[i:i + 1], ts, normalize=False)

    def edge_list(i):
        return seq_edge_list(i, sm["neighbor_edge_num"], sm["loop_edge_num"], sm["loop_dist_min"])[0]
    recs, book, _feats = replay(m, sm["nkf"], add_view, edge_list, float(g["thres"]), H, Wd, schedule=schedule)
    torch.cuda.synchronize()

    def n(t):
        return None if t is None else t.detach().cpu().numpy()
    edges = [dict(i=r.i, j=r.j, pose=n(r.pose), conf=float(r.rel_pose_conf), accepted=bool(r.accepted), confs=n(r.confs), intri=n(r.intri),
                  depths=n(r.depths), scales=[None if s is None else float(s) for s in r.scales],
                  scale_confs=[None if s is None else float(s) for s in r.scale_confs]) for r in recs]
    return edges, g, meta, m


@pytest.mark.parametrize("schedule", ["split", "batched", "pipelined"])
@pytest.mark.parametrize("case", SEQ_TINY + SEQ_FULL)
def test_keyframe_sequence_vs_reference_golden(G, case, schedule):
    """Every candidate edge of a >= 8-keyframe replay - pose, confidence, accept / reject decision, shared K, depth and
    confidence maps (lattice + off-lattice pixels + norms), and the scale edges between overlapping edges - against the
    reference, three ways: (a) "split" = the four split calls exactly as slam.py:142-189 issues them, B = 1 per edge (the
    zero-edit drop-in path); (b) "batched" = regress_views; (c) "pipelined" = the three-stream schedule bench.py's slam_replay
    reports (vista_slam_amd/keyframe_pipeline.py).  Decisions identical, numbers within 1e-3 in rel-L2 AND in max-abs."""
    from helpers import compare_seq_edges
    edges, g, meta, m = _seq_run(G, case, schedule)
    worst = compare_seq_edges(edges, g, meta, tol=TOL)
    acc = [e["accepted"] for e in edges]
    assert any(acc) and not all(acc)
    if not case.endswith("_t075"):
        assert any(e["i"] - e["j"] != 1 and e["accepted"] for e in edges), "no accepted non-adjacent edge in the fixture"
    assert m.range_report() == (0, 0)
    if case in SEQ_FULL:       # every scale edge of the full-architecture sequences is well conditioned: all of them met the bar
        assert worst.get("n_ill_conditioned", 0) == 0 and worst.get("n_well_conditioned", 0) > 0      # relative to the value itself
    print(f"[seq] {case} {schedule}: " + " ".join(f"{k}={v:.1e}" for k, v in sorted(worst.items())))


def test_keyframe_sequence_schedules_agree(G):
    """The three schedules against EACH OTHER on one sequence: same decisions, poses / maps / scale edges equal to the
    self-consistency bound of two valid schedules (different batch sizes take different tile families and K slices)."""
    import numpy as np
    from helpers import rel_l2
    case = "seq_tum_full_224"
    runs = {s: _seq_run(G, case, s)[0] for s in ("split", "batched", "pipelined")}
    ref = runs["batched"]
    for s in ("split", "pipelined"):
        for a, b in zip(runs[s], ref):
            assert (a["i"], a["j"], a["accepted"]) == (b["i"], b["j"], b["accepted"])
            assert rel_l2(a["pose"], b["pose"]) < ctol(DEFAULT)
            if a["accepted"]:
                assert rel_l2(a["depths"], b["depths"]) < ctol(DEFAULT) and rel_l2(a["confs"], b["confs"]) < ctol(DEFAULT)
    # the pipelined schedule runs the SAME batched calls on other streams: bit-identical to "batched"
    for a, b in zip(runs["pipelined"], ref):
        assert np.array_equal(a["pose"], b["pose"]) and (not a["accepted"] or np.array_equal(a["depths"], b["depths"]))


# ---------------------------------------------------------------------------------------------------------
# The documented integration path (slam.py:9,95-106): a checkpoint FILE -> torch.load ->
# STA().load_state_dict(checkpoint['model'], strict=True) -> .to(device) -> .eval(), reached through the reference's own import
# path with no edit to the caller (vista_slam_amd.install_as_reference).
_STANDIN_CALLER = '''"""Stand-in for the frontend half of the SLAM loop (vista_slam/slam.py needs pypose / cv2 / DBoW3Py): it imports the model class by
the reference's module path, builds it with no arguments, loads a checkpoint file strictly, moves it to the GPU, switches to eval
- the steps of load_frontend (slam.py:95-106) - and then uses the four split entry points like regress_two_views does."""
import torch
from .sta_model.sta_model import SymmetricTwoViewAssociation as STA


class Loop:
    def __init__(self, ckpt_path):
        self.device = torch.device("cuda")
        model = STA()
        weights = torch.load(ckpt_path, map_location="cpu", weights_only=False)["model"]
        model.load_state_dict(weights, strict=True)
        model.to(self.device)
        model.eval()
        self.frontend = model
        self.total_params = sum(t.numel() for t in model.parameters())

    def pair(self, img_i, img_j, shape):
        m = self.frontend
        with torch.no_grad():
            (fi, pi), (fj, pj) = (m._encode_image(x, shape, normalize=False) for x in (img_i, img_j))
            toks_i, toks_j = m._decode_stereo(fi, fj, pi, pj)
            pose = m.head_pose_s(toks_i[-1][:, 0, :])
            out_i = m.head_pts([fi] + [t[:, 1:, :].float() for t in toks_i], shape)
            out_j = m.head_pts([fj] + [t[:, 1:, :].float() for t in toks_j], shape)
        return pose, out_i, out_j
'''


def test_checkpoint_file_through_the_reference_import_path(G, tmp_path, monkeypatch):
    """torch.save({'model': state_dict}) with torch tensors in the checkpoint's key order (both alias keys of every shared DPT
    tensor included) -> a caller that imports `SymmetricTwoViewAssociation` by the reference's module path and loads the file
    exactly as slam.py:95-106 does -> outputs BIT-identical to the procedural-weights frontend every other test uses, and equal
    to the reference golden; the parameter count the SLAM loop prints (slam.py:46) is the architecture's."""
    import importlib
    import sys
    import numpy as np
    import torch
    import vista_slam_amd
    from helpers import load_golden, rel_l2
    from vista_slam_amd import weights as W
    sd = W.state_dict(W.FULL, seed=43)
    aliases = [k for k in sd if W._alias_of(k) != k]
    assert len(sd) == 665 and len(aliases) == 4 and all(W._alias_of(k) in sd for k in aliases)
    ckpt = tmp_path / "frontend_sta_weights.pth"
    torch.save({"model": {k: torch.from_numpy(v.copy()) for k, v in sd.items()}, "epoch": 0}, str(ckpt))
    del sd
    pkg = tmp_path / "vista_slam"
    pkg.mkdir()
    (pkg / "__init__.py").write_text("")
    (pkg / "loop.py").write_text(_STANDIN_CALLER)
    monkeypatch.syspath_prepend(str(tmp_path))
    for k in [k for k in sys.modules if k == "vista_slam" or k.startswith("vista_slam.")]:
        monkeypatch.delitem(sys.modules, k)
    vista_slam_amd.install_as_reference()
    loop = importlib.import_module("vista_slam.loop").Loop(str(ckpt))
    assert type(loop.frontend).__name__ == "STAFrontend" and loop.frontend._finalized
    ref_params = sum(int(np.prod(shape)) for name, shape, _k, _f in W.schema(W.FULL) if W._alias_of(name) == name)
    assert loop.total_params == ref_params
    g, meta = load_golden("full_224_b1")
    imgs = torch.from_numpy(W.synth_images(2, 224, 224, seed=43, tag=0)).cuda()
    shape = torch.tensor([[224, 224]])
    pose, ij, ji = loop.pair(imgs[:1], imgs[1:], shape)
    m = G.model("full", 1.0, DEFAULT)                                   # load_procedural: same values, never through a file
    fa, pa = m._encode_image(imgs[:1], shape, normalize=False)
    fb, pb = m._encode_image(imgs[1:], shape, normalize=False)
    d1, d2 = m._decode_stereo(fa, fb, pa, pb)
    pose2 = m.head_pose_s(d1[-1][:, 0, :])
    ij2 = m.head_pts([fa] + [t[:, 1:, :] for t in d1], shape)
    torch.cuda.synchronize()
    assert torch.equal(pose["pose"], pose2["pose"]) and torch.equal(ij["pts3d"], ij2["pts3d"]) and torch.equal(ij["conf"], ij2["conf"])
    sub = int(meta["sub"])
    assert rel_l2(ij["pts3d"].cpu().numpy()[:, ::sub, ::sub], g["main_pts3d"]) < TOL
    assert rel_l2(ji["pts3d"].cpu().numpy()[:, ::sub, ::sub], g["supp_pts3d"]) < TOL
    assert rel_l2(pose["pose"].cpu().numpy(), g["main_pose"]) < TOL
    for k in [k for k in sys.modules if k == "vista_slam" or k.startswith("vista_slam.")]:
        monkeypatch.delitem(sys.modules, k)


def test_checkpoint_with_fp16_tensors_and_shuffled_keys(G, tmp_path):
    """A checkpoint stored in fp16 (and in another key order) loads like the fp32 tensors of the same VALUES: load_state_dict
    converts every tensor to fp32 on the host (sta_frontend._load_one), the key order is irrelevant (slots are named)."""
    import numpy as np
    import torch
    from vista_slam_amd import weights as W
    from vista_slam_amd.sta_frontend import STAFrontend
    sd = W.state_dict(W.TINY, seed=43)
    half = {k: torch.from_numpy(v.copy()).half() for k, v in sd.items()}
    keys = list(half)
    rng = np.random.default_rng(3)
    rng.shuffle(keys)
    path = tmp_path / "tiny_fp16.pth"
    torch.save({"model": {k: half[k] for k in keys}}, str(path))
    ck = torch.load(str(path), map_location="cpu", weights_only=False)
    assert next(iter(ck["model"].values())).dtype == torch.float16
    a = STAFrontend(W.TINY, "cuda:0").load_state_dict(ck["model"], strict=True)
    b = STAFrontend(W.TINY, "cuda:0").load_state_dict({k: half[k].float().numpy() for k in half}, strict=True)
    imgs = torch.from_numpy(W.synth_images(2, 48, 64, seed=43, tag=0)).cuda()
    oa, ob = a.forward_pair(imgs[:1], imgs[1:]), b.forward_pair(imgs[:1], imgs[1:])
    torch.cuda.synchronize()
    for x, y in zip(oa, ob):
        for k in x:
            assert torch.equal(x[k], y[k]), k
    # a missing key and a foreign key still fail like the reference's strict load
    bad = dict(ck["model"]); bad.pop(keys[0])
    with pytest.raises(Exception, match="missing key"):
        STAFrontend(W.TINY, "cuda:0").load_state_dict(bad, strict=True)


def test_decode_stereo_positions_are_checked(G):
    """`_decode_stereo` and its positions argument.  RoPE is fused into the QKV epilogues and evaluated on the patch grid, so the shim
    has to KNOW whether a positions tensor is the grid: the tensors `_encode_image` returned pass through a provenance tag (no device
    sync), a foreign tensor with the same values is accepted after one comparison (cached), and since round 6 any OTHER positions
    are served by sta_decode_pos (the reference rotates by whatever it is handed; parity: test_decode_stereo_foreign_positions_*)."""
    import torch
    from vista_slam_amd import weights as W
    m = G.model("tiny", 1.0, DEFAULT)
    imgs = torch.from_numpy(W.synth_images(2, 48, 64, seed=43, tag=0)).cuda()
    fa, pa = m._encode_image(imgs[:1], None, normalize=False)
    fb, pb = m._encode_image(imgs[1:], None, normalize=False)
    assert getattr(pa, "_sta_grid", None)[:2] == (3, 4)
    ref1, ref2 = m._decode_stereo(fa, fb, pa, pb)
    got1, got2 = m._decode_stereo(fa, fb, pa.clone().cpu(), pb.clone())          # foreign tensors (no tag, one on the CPU): same grid -> same result
    torch.cuda.synchronize()
    assert torch.equal(ref1[-1], got1[-1]) and torch.equal(ref2[-1], got2[-1])
    sh1, _ = m._decode_stereo(fa, fb, pa + 1, pb)                                  # shifted positions: rotated by as given (sta_decode_pos)
    assert not torch.equal(sh1[-1], ref1[-1])
    with pytest.raises(AssertionError):
        m._decode_stereo(fa, fb, pa, pb[:, :6])                                    # another token count
    with pytest.raises(ValueError, match="below -1"):
        m._decode_stereo(fa, fb, pa - 2, pb)
    # the table form with the GRID's own positions = the grid form up to one more fp16-plane split of q / k (same angles, same table)
    import ctypes as C
    from vista_slam_amd import _lib
    L = m.cfg.dec_depth + 1
    o1 = [torch.empty(1, 13, m.cfg.dec_embed_dim, device="cuda") for _ in range(L)]
    o2 = [torch.empty(1, 13, m.cfg.dec_embed_dim, device="cuda") for _ in range(L)]
    p1 = (C.c_void_p * L)(*[t.data_ptr() for t in o1]); p2 = (C.c_void_p * L)(*[t.data_ptr() for t in o2])
    qa, qb = pa.contiguous(), pb.contiguous()
    _lib.check(m.lib.sta_decode_pos(m._h, fa.data_ptr(), fb.data_ptr(), qa.data_ptr(), qb.data_ptr(), 1, 12, 3, p1, p2, m._stream()))
    torch.cuda.synchronize()
    for a, b in zip(o1 + o2, list(ref1) + list(ref2)):
        assert float((a - b).norm() / b.norm()) < 2e-6 and not torch.equal(a, torch.zeros_like(a))
    # the provenance tag carries the tensor's version counter - an IN-PLACE edit of a tagged tensor keeps the Python
    # attribute but no longer passes as the patch grid: it is classified by its VALUES (here: a shifted window -> the table form)
    pc = m._encode_image(imgs[:1], None, normalize=False)[1]
    pc.add_(1)
    assert getattr(pc, "_sta_grid", None) is not None and m._grid_from_pos(pc, 12) == (None, None, 4)
    ed1, _ = m._decode_stereo(fa, fb, pc, pb)
    torch.cuda.synchronize()
    assert torch.equal(ed1[-1], sh1[-1])
    # ... and a verified foreign tensor is compared with the grid ONCE (cached by address / version / shape), not per call
    foreign = pa.clone()
    m._decode_stereo(fa, fb, foreign, pb)
    n0 = len(m._pos_verified)
    m._decode_stereo(fa, fb, foreign, pb)
    assert len(m._pos_verified) == n0 and any(v[3] is foreign for v in m._pos_verified.values())
    foreign.add_(1)                                                                # edited after it was verified: classified again
    assert m._grid_from_pos(foreign, 12) == (None, None, 4)


@pytest.mark.parametrize("prec", [DEFAULT, "f16x3"])
@pytest.mark.parametrize("case", ["decpos_tiny_48x64_b2", "decpos_tiny_48x80_sharp", "decpos_full_224_b1"])
def test_decode_stereo_foreign_positions_vs_reference_golden(G, case, prec):
    """VERDICT r5 'missing' item 4: `_decode_stereo` with positions other than the patch grid (`sta_blocks.py:134-137,196-199` rotate
    by whatever they are handed).  Fixtures from the reference (`gen_golden.gen_decpos`): windows of a larger grid (both views shifted
    differently), the second view's positions in reverse token order, another grid with the same token count; the HIP path
    (`sta_decode_pos` through the shim) against them on every hook layer of both sides, rel-L2 and max norm."""
    import numpy as np
    import torch
    from helpers import load_golden, rel_l2, max_rel
    from vista_slam_amd import weights as W
    g, meta = load_golden(case)
    full = case.startswith("decpos_full")
    if full:
        G.drop_models()
    cfg = W.FULL if full else W.TINY
    m = G.model("full" if full else "tiny", float(meta["qk_gain"]), prec, seed=int(meta["seed"]))
    m.range_report(reset=True)
    H, Wd, B, tsub = int(meta["H"]), int(meta["W"]), int(meta["B"]), int(meta["tsub"])
    imgs = torch.from_numpy(W.synth_images(2 * B, H, Wd, seed=int(meta["seed"]), tag=0)).cuda()
    fa, pa = m._encode_image(imgs[:B], None, normalize=False)
    fb, pb = m._encode_image(imgs[B:], None, normalize=False)
    if not full:        # tiny: the decoder alone, on the reference's own encoder features
        assert rel_l2(fa.cpu().numpy(), g["enc_feat_a"]) < TOL
        fa, fb = torch.from_numpy(g["enc_feat_a"]).cuda(), torch.from_numpy(g["enc_feat_b"]).cuda()
    last = cfg.hooks[-1] - 1
    errs = {}
    for tag in ("shift", "flip", "regrid"):
        qa, qb = torch.from_numpy(g[f"{tag}_pos_a"]), torch.from_numpy(g[f"{tag}_pos_b"]).cuda()      # one on the CPU, one on the device
        d1, d2 = m._decode_stereo(fa, fb, qa, qb)
        torch.cuda.synchronize()
        for hk in cfg.hooks[1:]:
            for side, d in (("dec1", d1), ("dec2", d2)):
                got, want = d[hk - 1].cpu().numpy()[:, ::tsub], g[f"{tag}_{side}_hook{hk - 1}"]
                errs[f"{tag}_{side}_hook{hk - 1}"] = max(rel_l2(got, want), max_rel(got, want))
        assert rel_l2(g["grid_dec1_last"], g[f"{tag}_dec1_hook{last}"]) > 3 * TOL, tag     # the positions matter (least at full depth with default-scale weights: 4e-3 ... 3e-2)
    print(case, prec, "worst", max(errs.values()))
    bad = {k: v for k, v in errs.items() if v > TOL}
    assert not bad, bad
    assert tuple(m.range_report(reset=True)) == (0, 0)


def test_reserve_then_no_allocation_and_no_device_sync(G):
    """SURVEY 8(b) "no hidden allocation per call" (VERDICT r5 item 4).  After sta_reserve(B, H, W, max_edges, streams) calls of at
    most those sizes on those streams neither allocate nor synchronise the device: the library's own counters (sta_alloc_stats:
    allocations / frees / stream + event creations, device-wide synchronisations of the compute entry points) and the device's
    free memory (hipMemGetInfo) are the same before and after forward_pair, the split entry points, both phases of the keyframe
    scheduler, and a FIRST call on a reserved stream the handle has never run on; an un-reserved larger shape still works and
    shows up in the counters."""
    import torch
    from vista_slam_amd import weights as W
    from vista_slam_amd.sta_frontend import STAFrontend
    from vista_slam_amd.slam_scheduler import regress_views, regress_views_begin, regress_views_finish
    m = STAFrontend(W.TINY, "cuda:0", precision=DEFAULT).load_procedural(seed=43)       # a fresh handle: no context, no workspace yet
    ref = G.model("tiny", 1.0, DEFAULT)
    H, Wd, B, k = 48, 64, 2, 3
    imgs = torch.from_numpy(W.synth_images(2 * B, H, Wd, seed=43, tag=5)).cuda()
    s1, s2, s3 = torch.cuda.Stream(), torch.cuda.Stream(), torch.cuda.Stream()
    # outputs / inputs of the calls below are allocated by torch: do that BEFORE the measurement window (torch's caching
    # allocator would otherwise move hipMemGetInfo by itself)
    want = ref.forward_pair(imgs[:B], imgs[B:])
    feats_ref = [ref._encode_image(imgs[v:v + 1], None, normalize=False)[0] for v in range(4)]
    edges_ref = regress_views(ref, feats_ref[3], feats_ref[:3], [False, False, True], -1.0, H, Wd)
    torch.cuda.synchronize()
    assert m.alloc_stats() == (0, 0) and m.workspace_bytes() == 0
    m.reserve(B, H, Wd, max_edges=k, streams=[torch.cuda.current_stream(), s1, s2, s3])
    a0 = m.alloc_stats()
    assert a0[0] > 0 and m.workspace_bytes() > 0
    m.reserve(B, H, Wd, max_edges=k, streams=[s1, s2])                                 # idempotent: nothing left to allocate
    assert m.alloc_stats() == a0

    def run_all(fe, stream):
        with torch.cuda.stream(stream):
            out = fe.forward_pair(imgs[:B], imgs[B:])
            feats = [fe._encode_image(imgs[v:v + 1], None, normalize=False)[0] for v in range(4)]
            p = regress_views_begin(fe, feats[3], feats[:3], H, Wd)
            got = regress_views_finish(fe, p, [False, False, True], -1.0)
            f1, p1 = fe._encode_image(imgs[:B], None, normalize=False)
            f2, p2 = fe._encode_image(imgs[B:], None, normalize=False)
            d1, d2 = fe._decode_stereo(f1, f2, p1, p2)
            fe.head_pose_s(d1[-1][:, 0, :])
        stream.synchronize()
        return out, got

    # torch's caching allocator keeps one pool PER STREAM and would move hipMemGetInfo by itself: warm the pools of s1 and s3 with
    # the same tensor sizes first - s3's through the OTHER handle, so that `m` itself has never run on s3
    run_all(m, s1)
    run_all(ref, s3)
    torch.cuda.synchronize()
    a1 = m.alloc_stats()
    assert a1 == a0, f"first calls after sta_reserve allocated / synchronised: {a0} -> {a1}"
    free0 = torch.cuda.mem_get_info()[0]
    out, got = run_all(m, s1)
    out3, got3 = run_all(m, s3)                           # s3: reserved, never used by this handle before - its FIRST calls
    torch.cuda.synchronize()
    assert m.alloc_stats() == a0
    assert torch.cuda.mem_get_info()[0] == free0, "device free memory moved across reserved calls"
    for o in (out, out3):
        assert torch.equal(o[0]["pts3d_pred"], want[0]["pts3d_pred"]) and torch.equal(o[1]["conf"], want[1]["conf"])
    for g in (got, got3):
        for a, b in zip(g, edges_ref):
            assert torch.equal(a.pose, b.pose) and torch.equal(a.depths, b.depths)
    # an un-reserved (larger) shape still works - lazily, and the counters say so
    big = torch.from_numpy(W.synth_images(2, 96, 128, seed=43, tag=6)).cuda()
    m.forward_pair(big[:1], big[1:])
    torch.cuda.synchronize()
    a2 = m.alloc_stats()
    assert a2[0] > a0[0] and a2[1] > a0[1]


def test_pipeline_streams_survive_a_larger_request(G):
    """sta_pipeline_streams(2) followed by sta_pipeline_streams(3) used to destroy and re-create every stream
    under the torch wrappers handed out by the first call.  Now the first call creates and probes all four and later calls return
    a prefix of the same list: the handles are stable and the first wrappers stay usable."""
    import torch
    from vista_slam_amd import weights as W
    m = G.model("tiny", 1.0, DEFAULT)
    imgs = torch.from_numpy(W.synth_images(2, 48, 64, seed=43, tag=7)).cuda()
    ref = m.forward_pair(imgs[:1], imgs[1:])[0]["pts3d_pred"].clone()
    two = m.pipeline_streams(2)
    three = m.pipeline_streams(3)
    four = m.pipeline_streams(4)
    assert [s.cuda_stream for s in two] == [s.cuda_stream for s in three[:2]] == [s.cuda_stream for s in four[:2]]
    assert three[2].cuda_stream == four[2].cuda_stream and len({s.cuda_stream for s in four}) == 4
    for s in two + four:
        with torch.cuda.stream(s):
            out = m.forward_pair(imgs[:1], imgs[1:])[0]["pts3d_pred"]
        s.synchronize()
        assert torch.equal(out, ref)


# ---------------------------------------------------------------------------------------------------------
# Real-checkpoint acceptance kit.  Fixtures: tests/golden/ckpt_*.npz +
# seq_*ckpt*.npz, written by `python oracle/gen_golden.py --checkpoint FILE` in the build container (the reference with the
# FILE's weights); the file itself: $STA_CHECKPOINT on the GPU box.  Neither exists today (pretrains/README.md:1-4: a download),
# so test_real_checkpoint_goldens skips - and test_checkpoint_kit_on_a_standin_checkpoint runs the SAME code path on a
# torch.save'd checkpoint holding the procedural weights, against the fixtures those weights have (bit-equal to what the generator
# writes for that file: oracle/check_oracle_vs_ref.py ckpt).
def _checkpoint_frontend(path):
    import torch
    from vista_slam_amd import weights as W
    from vista_slam_amd.sta_frontend import STAFrontend
    ck = torch.load(path, map_location="cpu", weights_only=False)          # slam.py:97-100
    sd = ck["model"] if isinstance(ck, dict) and "model" in ck else ck
    fe = STAFrontend(W.FULL, "cuda:0").load_state_dict(sd, strict=True)
    return fe, W.state_dict_fingerprint(sd)


def _checkpoint_acceptance(G, path, fwd_cases, seq_cases, precisions=(DEFAULT, "f16x3")):
    """Every forward fixture in both precisions (all outputs, split entry points, taps; rel-L2 and max norm at 1e-3, range
    report (0, 0)) and every sequence fixture under the three schedules, on a frontend whose weights came from `path`."""
    from helpers import compare_seq_edges, load_golden
    fe, fp = _checkpoint_frontend(path)
    report = []
    for name in fwd_cases:
        g, _meta = load_golden(name)
        if "ckpt_fingerprint" in g:
            assert str(g["ckpt_fingerprint"]) == fp, f"{name} was generated from another checkpoint than {path}"
        for prec in precisions:
            r = G.run_golden_case(name, prec, frontend=fe)
            bad = {k: v for k, v in r.items() if not v < TOL}
            assert not bad, (name, prec, bad)
            assert G.last_range == (0, 0), (name, prec, G.last_range)
            report.append(f"{name} {prec}: worst {max(r.values()):.2e} ({max(r, key=r.get)})")
    for name in seq_cases:
        g, _meta = load_golden(name)
        if "ckpt_fingerprint" in g:
            assert str(g["ckpt_fingerprint"]) == fp, f"{name} was generated from another checkpoint than {path}"
        for schedule in ("split", "batched", "pipelined"):
            edges, g, meta, _m = _seq_run(G, name, schedule, frontend=fe)
            worst = compare_seq_edges(edges, g, meta, tol=TOL)
            assert fe.range_report() == (0, 0)
            report.append(f"{name} {schedule}: " + " ".join(f"{k}={v:.1e}" for k, v in sorted(worst.items())))
    del fe
    return report


def test_real_checkpoint_goldens(G):
    """$STA_CHECKPOINT + tests/golden/ckpt_*.npz: parity on REAL weights.  Skips while either is absent."""
    import glob
    import os
    path = os.environ.get("STA_CHECKPOINT", "")
    gold = os.path.join(os.path.dirname(__file__), "golden")
    fwd = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(gold, "ckpt_*.npz")))
    seq = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(gold, "seq_*ckpt*.npz")))
    if not path or not os.path.exists(path):
        pytest.skip("no real checkpoint on this box ($STA_CHECKPOINT unset): pretrains/frontend_sta_weights.pth is a download")
    if not fwd:
        pytest.skip("no tests/golden/ckpt_*.npz: run `python oracle/gen_golden.py --checkpoint $STA_CHECKPOINT` in the build container first")
    G.drop_models()
    for line in _checkpoint_acceptance(G, path, fwd, seq):
        print("[ckpt]", line)


def test_checkpoint_kit_on_a_standin_checkpoint(G, tmp_path):
    """The same acceptance path on a stand-in FILE: the procedural full-architecture weights torch.save'd as {'model': ...} (both
    alias keys of the shared DPT tensors, like the real file) against the fixtures of those weights - full_224_b1 (= what the
    generator names ckpt_224_b1) and seq_tum_full_224_t075 (= seq_tum_ckpt_224: the yaml's own rel_pose_thres 0.75)."""
    import torch
    from vista_slam_amd import weights as W
    sd = W.state_dict(W.FULL, seed=43)
    path = tmp_path / "standin_frontend_sta_weights.pth"
    torch.save({"model": {k: torch.from_numpy(v.copy()) for k, v in sd.items()}, "epoch": 0}, str(path))
    del sd
    G.drop_models()
    report = _checkpoint_acceptance(G, str(path), ["full_224_b1"], ["seq_tum_full_224_t075"], precisions=(DEFAULT,))
    for line in report:
        print("[ckpt stand-in]", line)
    assert len(report) == 4


@pytest.mark.parametrize("case", ["full_224_b1", "full_224_b1_sharp", "full_224_b1_outlier", "full_384x512_b1_sharp"])
def test_opt_in_precision_f16x3m_full_architecture(G, case):
    """Precision f16x3m (round 6: f16x3h + mlp.fc2 of both transformers in the f16mx arithmetic, +2.1 % at the headline configuration)
    is NOT the default: the written rule of DESIGN.md section 2 - every committed golden <= 0.5 x the bar in rel-L2 and < the bar
    in the max norm - fails on two of the ten tiny-configuration stress sets (3.4e-3, 7.4e-4 / 1.0e-3).  What the opt-in mode does
    hold is asserted here: every full-architecture golden (default, sharp, outlier statistics; 224x224 and the headline resolution)
    within HALF the bar in both norms, no range event."""
    r = G.run_golden_case(case, "f16x3m")
    bad = {k: v for k, v in r.items() if not v < 0.5 * TOL}
    assert not bad, bad
    assert G.last_range == (0, 0)


@pytest.mark.parametrize("case", ["full_384x512_b1_sharp", "full_384x512_b1_outlier"])
def test_batch8_stress_weights_on_the_throughput_kernels(G, case):
    """The B = 1 goldens of the stress weights run the small-grid tile family in the transformer (M = 1536 rows); the kernels the
    BENCHMARK runs - 192x128 / 192x256 / 256x256 GEMM families, the paired QKV launch, the halo convolutions with the fused DPT tail
    on transposed accumulators - had only ever seen default-scale weights at B = 8.  Here: 8 pairs @512x384 with peaky attention
    (Q/K gain 3) and with the checkpoint-like outlier statistics; slot 0 holds the golden's pair (outputs and encoder features
    against the reference), slot 7 must equal the same pair run alone."""
    import numpy as np
    import torch
    from helpers import load_golden, rel_l2, max_rel
    from vista_slam_amd import weights as W
    g, meta = load_golden(case)
    G.drop_models()
    m = G.model("full", float(meta["qk_gain"]), DEFAULT, int(meta["seed"]), int(meta.get("outlier", 0)))
    m.range_report(reset=True)
    H, Wd, sub, B = int(meta["H"]), int(meta["W"]), int(meta["sub"]), 8
    im = W.synth_images(2, H, Wd, seed=int(meta["seed"]), tag=0)
    extra = W.synth_images(2 * B, H, Wd, seed=int(meta["seed"]), tag=5)
    a = torch.from_numpy(np.concatenate([im[:1], extra[:B - 1]])).cuda()
    b = torch.from_numpy(np.concatenate([im[1:], extra[B:2 * B - 1]])).cuda()
    main, supp = m.forward_pair(a, b)
    fa, _pa = m._encode_image(a, None, normalize=False)
    torch.cuda.synchronize()
    errs = {}
    for side, o in (("main", main), ("supp", supp)):
        pts, conf = o["pts3d_pred"][:1].cpu().numpy(), o["conf"][:1].cpu().numpy()
        errs[f"{side}_pts3d"] = rel_l2(pts[:, ::sub, ::sub], g[f"{side}_pts3d"])
        errs[f"{side}_pts3d_max"] = max_rel(pts[:, ::sub, ::sub], g[f"{side}_pts3d"])
        errs[f"{side}_conf"] = rel_l2(conf[:, ::sub, ::sub], g[f"{side}_conf"])
        errs[f"{side}_pts3d_rand"] = rel_l2(pts.reshape(1, -1, 3)[:, g["rand_idx"]], g[f"{side}_pts3d_rand"])
        errs[f"{side}_pose"] = rel_l2(o["relative_pose"][:1].cpu().numpy(), g[f"{side}_pose"])
        errs[f"{side}_pose_conf"] = rel_l2(o["relative_pose_conf"][:1].cpu().numpy(), g[f"{side}_pose_conf"])
    errs["enc_feat_a"] = rel_l2(fa[:1].cpu().numpy()[:, ::sub], g["enc_feat_a"])
    bad = {k: v for k, v in errs.items() if not v < TOL}
    assert not bad, bad
    # regression guard at twice what the default policy measures here (7.5e-5 outlier, 3.6e-5 sharp): this test caught head.4's
    # weights - 1e-5 under the outlier statistics - entering the fused tail as fp16 SUBNORMALS (3.4e-4); they are now split after
    # an exact power-of-two scaling per output row (sta_finalize_weights, head_epilogue_t)
    assert errs["main_pts3d"] < 1.5e-4 and errs["supp_pts3d"] < 1.5e-4, errs
    assert m.range_report() == (0, 0)
    i = B - 1
    m1, s1 = m.forward_pair(a[i:i + 1], b[i:i + 1])
    torch.cuda.synchronize()
    # (two schedules of the same forward on weights that amplify rounding differences: a tenth of the bar)
    assert rel_l2(m1["pts3d_pred"].cpu().numpy(), main["pts3d_pred"][i:i + 1].cpu().numpy()) < 1e-4
    assert rel_l2(s1["relative_pose"].cpu().numpy(), supp["relative_pose"][i:i + 1].cpu().numpy()) < 1e-4
    print(f"[b8 stress] {case}: " + " ".join(f"{k}={v:.1e}" for k, v in sorted(errs.items())))


def test_tiny_weight_tensor_is_reported(G):
    """The low end of the fp16 range: a packed weight tensor whose every value is below 2^-12 (its operand planes would be fp16
    subnormals) shows up in the range report - permanently, as a property of the loaded weights - and disappears when the slot is
    loaded with ordinary values again; an all-zero tensor is not an event."""
    import numpy as np
    from vista_slam_amd import weights as W
    from vista_slam_amd.sta_frontend import STAFrontend
    sd = W.state_dict(W.TINY, seed=43)
    key = "enc_blocks.0.mlp.fc2.weight"
    m = STAFrontend(W.TINY, "cuda:0").load_state_dict(sd, strict=True)
    assert m.range_report() == (0, 0)
    m._load_one(key, sd[key] * np.float32(1e-6))
    assert m.range_report() == (1, 0) and m.range_report() == (1, 0)          # survives the reset
    m._load_one(key, np.zeros_like(sd[key]))
    assert m.range_report() == (0, 0)
    m._load_one(key, sd[key] * np.float32(1e-6))
    m._load_one(key, sd[key])
    assert m.range_report() == (0, 0)


def test_fused_tail_implicit_gemm_form_matches_halo_form(G):
    """The fused DPT tail exists on two kernels - the halo-tiled convolution (the product's choice wherever it is legal) and the
    implicit-GEMM 192x128 tile (experiment switch 0 at >= 2M pixels; tools/ab_option.py 0 0 1) - both on transposed accumulators
    with head.4 on the matrix pipe (head_epilogue_t).  The cost model leaves the second one unused in the product flow, so it is
    exercised here: 8 pairs @512x384, outputs equal to the halo form's up to summation order, slot 0 against the reference golden."""
    import numpy as np
    import torch
    from helpers import load_golden, rel_l2
    from vista_slam_amd import _lib, weights as W
    g, meta = load_golden("full_384x512_b1")
    G.drop_models()
    m = G.model("full", 1.0, DEFAULT, hooks=True)
    sub, B = int(meta["sub"]), 8
    im = W.synth_images(2, 384, 512, seed=43, tag=0)
    extra = W.synth_images(2 * B, 384, 512, seed=43, tag=5)
    a = torch.from_numpy(np.concatenate([im[:1], extra[:B - 1]])).cuda()
    b = torch.from_numpy(np.concatenate([im[1:], extra[B:2 * B - 1]])).cuda()
    outs = []
    for sw in (0, 1):
        _lib.check(m.lib.sta_debug_set_option(m._h, 0, sw))
        main, supp = m.forward_pair(a, b)
        torch.cuda.synchronize()
        outs.append((main["pts3d_pred"].clone(), supp["conf"].clone()))
    _lib.check(m.lib.sta_debug_set_option(m._h, 0, 0))
    assert not torch.equal(outs[0][0], outs[1][0]), "the switch did not change the kernel"
    assert rel_l2(outs[1][0].cpu().numpy(), outs[0][0].cpu().numpy()) < 1e-6 and rel_l2(outs[1][1].cpu().numpy(), outs[0][1].cpu().numpy()) < 1e-6
    assert rel_l2(outs[1][0][:1].cpu().numpy()[:, ::sub, ::sub], g["main_pts3d"]) < TOL
    assert m.range_report() == (0, 0)
