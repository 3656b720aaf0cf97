"""GPU parity checks, shared by the pytest suite (tests/test_gpu_*.py) and the diagnostic runner
(tests/gpu_diag.py).  Every check calls the HIP product kernels through the C ABI and compares
with a plain fp32 torch/numpy reference of the same op (or with reference-derived goldens) and
returns {metric_name: error}.  Nothing here runs the product on a CPU fallback: there is none."""
import ctypes as C

import numpy as np
import torch

from helpers import load_golden, rel_l2, max_rel, rope2d_ref, grid_pos
from vista_slam_amd import _lib
from vista_slam_amd import weights as W
from vista_slam_amd.sta_frontend import STAFrontend, rope2d_inplace

DEV = "cuda:0"
_models = {}
last_range = (0, 0)      # (fp16 saturations, fp8 correction-byte saturations) counted during the last run_golden_case (sta_range_report)


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


def st():
    return torch.cuda.current_stream().cuda_stream


def model(cfg_name="tiny", qk_gain=1.0, precision="f16x3", seed=43, outlier=0, hooks=False):
    """Cached frontends (weights are procedural, so (cfg, gain, seed, outlier level) identifies them).  hooks=False: the PRODUCT
    library libsta_mi355.so (what every golden / parity test runs); hooks=True: the test-hooks build libsta_mi355_test.so (same
    translation unit + the kernel-level entry points and experiment switches of include/sta_mi355_debug.h)."""
    key = (cfg_name, qk_gain, seed, outlier, bool(hooks))
    if key not in _models:
        cfg = W.TINY if cfg_name == "tiny" else W.FULL
        m = STAFrontend(cfg, DEV, precision=precision, lib=_lib.load_test() if hooks else None)
        m.load_procedural(seed=seed, qk_gain=qk_gain, outlier=outlier)
        _models[key] = m
    m = _models[key]
    m.set_precision(precision)
    return m


def drop_models():
    _models.clear()
    torch.cuda.empty_cache()


def kernel_handle(precision, variant=0):
    """precision "head_mx": the DPT head's arithmetic of the default policy (f16 main product + one block-scaled fp8 correction
    MFMA on f16mx rows) in the kernels that have it: the debug GEMM (plane epilogue), conv3x3, ConvT, bilinear."""
    # "mlp_mx": the MLP's f16mx path of precision f16x3m - mlp.fc1's GELU epilogue writing f16mx rows (via_f16 + GELU) and mlp.fc2
    # (fp32 / in-place-residual epilogues on f16mx rows and weights)
    m = model("tiny", 1.0, {"head_mx": "f16x3h", "mlp_mx": "f16x3m"}.get(precision, precision), hooks=True)
    _lib.check(m.lib.sta_set_gemm_variant(m._h, variant))
    _lib.check(m.lib.sta_debug_set_option(m._h, 4, {"head_mx": 1, "mlp_mx": 2}.get(precision, 0)))
    return m, m.lib, m._h


def has_hooks(m):
    return hasattr(m.lib, "sta_set_gemm_variant")


def set_variant(m, variant):
    """Force a GEMM tile family (test-hooks build only).  On a product-library frontend only `0` (= what it always does) is legal."""
    if not has_hooks(m):
        assert variant == 0, "forced tile families need a frontend on the test-hooks library (model(..., hooks=True))"
        return
    _lib.check(m.lib.sta_set_gemm_variant(m._h, variant))


# ------------------------------------------------------------------------------------------ kernels
def check_gemm(precision, M=300, N=200, K=96, act=0, via_f16=0, resid=False, seed=0, variant=0):
    m, lib, h = kernel_handle(precision, variant)
    g = torch.Generator().manual_seed(seed)
    A = torch.randn(M, K, generator=g) * 1.3
    Wt = torch.randn(N, K, generator=g) * 0.1
    b = torch.randn(N, generator=g)
    R = torch.randn(M, N, generator=g) if resid else None
    ref = A.double() @ Wt.double().T + b.double()
    if act == 1:
        ref = torch.nn.functional.gelu(ref)
    elif act == 2:
        ref = torch.relu(ref)
    if resid:
        ref = ref + R.double()
    out = torch.empty(M, N, device=DEV)
    Ad, Wd, bd = A.to(DEV), Wt.to(DEV), b.to(DEV)
    Rd = R.to(DEV) if resid else None
    _lib.check(lib.sta_debug_gemm(h, Ad.data_ptr(), Wd.data_ptr(), bd.data_ptr(), M, N, K, act, via_f16,
                                  Rd.data_ptr() if resid else None, out.data_ptr(), st()))
    torch.cuda.synchronize()
    return {"rel_l2": rel_l2(out.cpu().numpy(), ref.numpy()), "max_rel": max_rel(out.cpu().numpy(), ref.numpy())}


def check_qkv_rope(precision, S=2, hp=3, wp=4, pose_tok=1, K=128, Cdim=128, seed=1, variant=0):
    m, lib, h = kernel_handle(precision, variant)
    g = torch.Generator().manual_seed(seed)
    ntok = hp * wp + pose_tok
    x = torch.randn(S * ntok, K, generator=g)
    Wt = torch.randn(3 * Cdim, K, generator=g) * 0.1
    b = torch.randn(3 * Cdim, generator=g) * 0.1
    heads = Cdim // 64
    npad = (ntok + 63) // 64 * 64
    q = torch.empty(S, heads, ntok, 64, device=DEV)
    k = torch.empty_like(q)
    vt = torch.empty(S * heads * 64, npad, device=DEV)
    xd, Wd, bd = x.to(DEV), Wt.to(DEV), b.to(DEV)      # keep device inputs alive across the call
    _lib.check(lib.sta_debug_qkv_rope(h, xd.data_ptr(), Wd.data_ptr(), bd.data_ptr(), S, ntok, K, Cdim,
                                      wp, pose_tok, q.data_ptr(), k.data_ptr(), vt.data_ptr(), st()))
    torch.cuda.synchronize()
    y = (x.double() @ Wt.double().T + b.double()).float().reshape(S, ntok, 3, heads, 64).permute(2, 0, 3, 1, 4).numpy()
    pos = grid_pos(S, hp, wp, pose_tok=bool(pose_tok))
    qr, kr, vr = rope2d_ref(y[0], pos), rope2d_ref(y[1], pos), y[2]
    v = vt.cpu().numpy().reshape(S, heads, 64, npad)[..., :ntok].transpose(0, 1, 3, 2)
    pad = vt.cpu().numpy().reshape(S, heads, 64, npad)[..., ntok:]
    return {"q": max_rel(q.cpu().numpy(), qr), "k": max_rel(k.cpu().numpy(), kr), "v": max_rel(v, vr),
            "vpad_abs": float(np.abs(pad).max()) if pad.size else 0.0}


def check_attention(precision, S=2, heads=2, nq=197, nk=197, kv_shift=0, sharp=1.0, seed=2):
    m, lib, h = kernel_handle(precision)
    g = torch.Generator().manual_seed(seed)
    q = torch.randn(S, heads, nq, 64, generator=g) * sharp
    k = torch.randn(S, heads, nk, 64, generator=g)
    v = torch.randn(S, heads, nk, 64, generator=g)
    idx = [(s + kv_shift) % S for s in range(S)]
    a = (q.double() @ k[idx].double().transpose(-1, -2)) * 0.125
    ref = (a.softmax(-1) @ v[idx].double()).permute(0, 2, 1, 3).reshape(S, nq, heads * 64)
    out = torch.empty(S, nq, heads * 64, device=DEV)
    qd, kd, vd = q.to(DEV), k.to(DEV), v.to(DEV)
    _lib.check(lib.sta_debug_attention(h, qd.data_ptr(), kd.data_ptr(), vd.data_ptr(), S, heads, nq, nk,
                                       kv_shift, out.data_ptr(), st()))
    torch.cuda.synchronize()
    o = out.cpu().numpy()
    return {"rel_l2": rel_l2(o, ref.numpy()), "max_rel": max_rel(o, ref.numpy()), "nan": float(np.isnan(o).sum())}


def check_gemm_tail(precision, tiles_m=8, tail=16, N=2304, K=256, act=0, via_f16=0, resid=False, variant=0, seed=11):
    """Dense GEMM whose last `tail` rows run on the skinny tail blocks (GemmParams::m_tail, the decoder's pose-token
    rows): M = tiles_m x 192 (or 256) + tail at a size where the throughput families are selected."""
    m, lib, h = kernel_handle(precision, variant)
    bm = 256 if variant == 2 else 192
    M = tiles_m * bm + tail
    _lib.check(lib.sta_debug_set_tail_hint(h, tail))
    try:
        r = check_gemm(precision, M=M, N=N, K=K, act=act, via_f16=via_f16, resid=resid, seed=seed, variant=variant)
    finally:
        _lib.check(lib.sta_debug_set_tail_hint(h, 0))
    return r


def check_qkv_rope_decoder_rows(precision, S=2, hp=3, wp=4, K=128, Cdim=128, seed=12, variant=0):
    """QKV + RoPE epilogue on the decoder's row order: x = [S*N patch rows | S pose rows]; the buffers hold N + 1 tokens per
    sequence with the pose token (position -1) last."""
    m, lib, h = kernel_handle(precision, variant)
    g = torch.Generator().manual_seed(seed)
    N = hp * wp
    ntok = N + 1
    xs = torch.randn(S, ntok, K, generator=g)                 # reference order: pose token first
    Wt = torch.randn(3 * Cdim, K, generator=g) * 0.1
    b = torch.randn(3 * Cdim, generator=g) * 0.1
    heads = Cdim // 64
    npad = (ntok + 63) // 64 * 64
    x_dec = torch.cat([xs[:, 1:].reshape(S * N, K), xs[:, 0]], 0).contiguous()
    q = torch.empty(S, heads, ntok, 64, device=DEV)
    k = torch.empty_like(q)
    vt = torch.empty(S * heads * 64, npad, device=DEV)
    xd, Wd, bd = x_dec.to(DEV), Wt.to(DEV), b.to(DEV)
    _lib.check(lib.sta_debug_qkv_rope(h, xd.data_ptr(), Wd.data_ptr(), bd.data_ptr(), S, N, K, Cdim,
                                      wp, 2, q.data_ptr(), k.data_ptr(), vt.data_ptr(), st()))
    torch.cuda.synchronize()
    y = (xs.reshape(S * ntok, K).double() @ Wt.double().T + b.double()).float().reshape(S, ntok, 3, heads, 64).permute(2, 0, 3, 1, 4).numpy()
    pos = grid_pos(S, hp, wp, pose_tok=True)
    qr, kr, vr = rope2d_ref(y[0], pos), rope2d_ref(y[1], pos), y[2]
    order = list(range(1, ntok)) + [0]                          # device token order: patches, then the pose token
    v = vt.cpu().numpy().reshape(S, heads, 64, npad)[..., :ntok].transpose(0, 1, 3, 2)
    pad = vt.cpu().numpy().reshape(S, heads, 64, npad)[..., ntok:]
    return {"q": max_rel(q.cpu().numpy(), qr[:, :, order]), "k": max_rel(k.cpu().numpy(), kr[:, :, order]), "v": max_rel(v, vr[:, :, order]),
            "vpad_abs": float(np.abs(pad).max()) if pad.size else 0.0}


def check_attention_pose(precision, S=2, heads=2, n=196, kv_shift=0, sharp=1.0, seed=13):
    """Decoder form of the attention kernel: n patch tokens + the pose token (last): as a key it is folded into the initial
    softmax state; as a query it rides in the last query block's spare rows (n % 128 != 0) or is served by the pose blocks."""
    m, lib, h = kernel_handle(precision)
    g = torch.Generator().manual_seed(seed)
    nt = n + 1
    q = torch.randn(S, heads, nt, 64, generator=g) * sharp
    k = torch.randn(S, heads, nt, 64, generator=g)
    v = torch.randn(S, heads, nt, 64, generator=g)
    idx = [(s + kv_shift) % S for s in range(S)]
    a = (q.double() @ k[idx].double().transpose(-1, -2)) * 0.125
    ref = (a.softmax(-1) @ v[idx].double()).permute(0, 2, 1, 3).reshape(S, nt, heads * 64)
    ref = torch.cat([ref[:, :n].reshape(S * n, heads * 64), ref[:, n]], 0)       # decoder row order
    out = torch.empty(S * n + S, heads * 64, device=DEV)
    qd, kd, vd = q.to(DEV), k.to(DEV), v.to(DEV)
    _lib.check(lib.sta_debug_attention_pose(h, qd.data_ptr(), kd.data_ptr(), vd.data_ptr(), S, heads, n, kv_shift, out.data_ptr(), st()))
    torch.cuda.synchronize()
    o = out.cpu().numpy()
    return {"rel_l2": rel_l2(o, ref.numpy()), "rel_l2_pose": rel_l2(o[S * n:], ref.numpy()[S * n:]),
            "max_rel": max_rel(o, ref.numpy()), "nan": float(np.isnan(o).sum())}


def check_conv3(precision, n=2, H=7, W_=5, Cin=32, Co=48, stride=1, relu_in=0, act=0, resid=False, seed=3, variant=0):
    m, lib, h = kernel_handle(precision, variant)
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(n, Cin, H, W_, generator=g)
    w = torch.randn(Co, Cin, 3, 3, generator=g) * 0.1
    b = torch.randn(Co, generator=g)
    xin = torch.relu(x) if relu_in else x
    ref = torch.nn.functional.conv2d(xin.double(), w.double(), b.double(), stride=stride, padding=1)
    if act == 2:
        ref = torch.relu(ref)
    R = torch.randn(ref.shape, generator=g) if resid else None
    if resid:
        ref = ref + R.double()
    Ho, Wo = ref.shape[2], ref.shape[3]
    out = torch.empty(n, Ho, Wo, Co, device=DEV)
    xd = x.permute(0, 2, 3, 1).contiguous().to(DEV)
    Rd = R.permute(0, 2, 3, 1).contiguous().to(DEV) if resid else None
    wd, bd = w.to(DEV), b.to(DEV)
    _lib.check(lib.sta_debug_conv3x3(h, xd.data_ptr(), wd.data_ptr(), bd.data_ptr(), n, H, W_, Cin, Co, stride,
                                     relu_in, act, Rd.data_ptr() if resid else None, out.data_ptr(), st()))
    torch.cuda.synchronize()
    o = out.cpu().permute(0, 3, 1, 2).numpy()
    return {"rel_l2": rel_l2(o, ref.numpy()), "max_rel": max_rel(o, ref.numpy())}


def check_convt(precision, n=2, H=3, W_=5, Cdim=96, k=4, seed=4, variant=0):
    m, lib, h = kernel_handle(precision, variant)
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(n, Cdim, H, W_, generator=g)
    w = torch.randn(Cdim, Cdim, k, k, generator=g) * 0.1
    b = torch.randn(Cdim, generator=g)
    ref = torch.nn.functional.conv_transpose2d(x.double(), w.double(), b.double(), stride=k)
    out = torch.empty(n, H * k, W_ * k, Cdim, device=DEV)
    xd, wd, bd = x.permute(0, 2, 3, 1).contiguous().to(DEV), w.to(DEV), b.to(DEV)
    _lib.check(lib.sta_debug_convt(h, xd.data_ptr(), wd.data_ptr(), bd.data_ptr(), n, H, W_, Cdim, k, out.data_ptr(), st()))
    torch.cuda.synchronize()
    o = out.cpu().permute(0, 3, 1, 2).numpy()
    return {"rel_l2": rel_l2(o, ref.numpy()), "max_rel": max_rel(o, ref.numpy())}


def check_up2(precision, n=2, H=7, W_=5, Cdim=16, crop=None, seed=5):
    m, lib, h = kernel_handle(precision)
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(n, Cdim, H, W_, generator=g)
    ref = torch.nn.functional.interpolate(x.double(), scale_factor=2, mode="bilinear", align_corners=True)
    Hc, Wc = crop if crop else (2 * H, 2 * W_)
    ref = ref[:, :, :Hc, :Wc]
    out = torch.empty(n, Hc, Wc, Cdim, device=DEV)
    xd = x.permute(0, 2, 3, 1).contiguous().to(DEV)
    _lib.check(lib.sta_debug_up2(h, xd.data_ptr(), n, H, W_, Cdim, Hc, Wc, out.data_ptr(), st()))
    torch.cuda.synchronize()
    o = out.cpu().permute(0, 3, 1, 2).numpy()
    return {"rel_l2": rel_l2(o, ref.numpy()), "max_rel": max_rel(o, ref.numpy())}


def check_layernorm(precision, M=37, Cdim=768, seed=6):
    m, lib, h = kernel_handle(precision)
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(M, Cdim, generator=g) * 3 + 0.7
    w = torch.randn(Cdim, generator=g)
    b = torch.randn(Cdim, generator=g)
    ref = torch.nn.functional.layer_norm(x.double(), (Cdim,), w.double(), b.double(), eps=1e-6).numpy()
    o32 = torch.empty(M, Cdim, device=DEV)
    op = torch.empty(M, Cdim, device=DEV)
    xd, wd, bd = x.to(DEV), w.to(DEV), b.to(DEV)
    _lib.check(lib.sta_debug_layernorm(h, xd.data_ptr(), wd.data_ptr(), bd.data_ptr(), M, Cdim, 1e-6,
                                       o32.data_ptr(), op.data_ptr(), st()))
    torch.cuda.synchronize()
    return {"f32": max_rel(o32.cpu().numpy(), ref), "planes": max_rel(op.cpu().numpy(), ref)}


def check_ops_golden(precision):
    """Single-op vectors produced by the reference's own modules (tests/golden/ops.npz)."""
    m, lib, h = kernel_handle(precision)
    g, _ = load_golden("ops")
    res = {}
    # RoPE2D in place (curope drop-in), tokens given as (B,H,N,D) -> kernel layout (B,N,H,D)
    tok = dev(g["rope_tok"]).permute(0, 2, 1, 3).contiguous()
    rpos = dev(g["rope_pos"])
    rope2d_inplace(tok, rpos, 100.0, 1.0)
    res["rope2d"] = max_rel(tok.permute(0, 2, 1, 3).cpu().numpy(), g["rope_out"])
    # inverse rotation (fwd = -1) restores the input (curope backward, curope2d.py:24-29)
    rope2d_inplace(tok, rpos, 100.0, -1.0)
    res["rope2d_roundtrip"] = max_rel(tok.permute(0, 2, 1, 3).cpu().numpy(), g["rope_tok"])
    # the other token dtypes curope dispatches on (kernels.cu:101): fp32 rotation of the stored value, result stored in that dtype
    for dt, name in ((torch.float16, "f16"), (torch.float64, "f64")):
        t0 = dev(g["rope_tok"]).permute(0, 2, 1, 3).contiguous().to(dt)
        want = torch.from_numpy(rope2d_ref(t0.float().permute(0, 2, 1, 3).cpu().numpy(), g["rope_pos"])).to(dt)
        rope2d_inplace(t0, rpos, 100.0, 1.0)
        res["rope2d_" + name] = max_rel(t0.permute(0, 2, 1, 3).cpu().double().numpy(), want.double().numpy())
    # LayerNorm eps 1e-6
    M, Cd = g["ln_x"].shape
    o32 = torch.empty(M, Cd, device=DEV); op = torch.empty(M, Cd, device=DEV)
    lx, lw, lb = dev(g["ln_x"]), dev(g["ln_w"]), dev(g["ln_b"])
    _lib.check(lib.sta_debug_layernorm(h, lx.data_ptr(), lw.data_ptr(), lb.data_ptr(),
                                       M, Cd, 1e-6, o32.data_ptr(), op.data_ptr(), st()))
    res["layernorm"] = max_rel(o32.cpu().numpy(), g["ln_out"])
    # SVD orthogonalisation incl. reflection / near-singular inputs
    B = g["svd_in"].shape[0]
    r = torch.empty(B, 3, 3, device=DEV)
    sv = dev(g["svd_in"])
    _lib.check(lib.sta_debug_svd_orthogonalize(h, sv.data_ptr(), r.data_ptr(), B, st()))
    res["svd_orth"] = float(np.abs(r.cpu().numpy() - g["svd_out"]).max())
    # bilinear x2 align_corners, odd size: channels padded to 8
    x = g["bilin_x"]
    xp = np.zeros((1, 8, x.shape[2], x.shape[3]), np.float32); xp[:, :3] = x
    out = torch.empty(1, 2 * x.shape[2], 2 * x.shape[3], 8, device=DEV)
    xpd = dev(xp.transpose(0, 2, 3, 1))
    _lib.check(lib.sta_debug_up2(h, xpd.data_ptr(), 1, x.shape[2], x.shape[3], 8,
                                 2 * x.shape[2], 2 * x.shape[3], out.data_ptr(), st()))
    res["bilinear"] = max_rel(out.cpu().numpy().transpose(0, 3, 1, 2)[:, :3], g["bilin_out"])
    # postprocess through head_final: weights = identity on the first 4 channels
    pin = g["post_in"]                       # [1,4,h,w]
    npix = pin.shape[2] * pin.shape[3]
    feat = np.zeros((npix, 128), np.float32); feat[:, :4] = pin[0].reshape(4, npix).T
    w4 = np.zeros((4, 128), np.float32); w4[np.arange(4), np.arange(4)] = 1.0
    pts = torch.empty(npix, 3, device=DEV); conf = torch.empty(npix, device=DEV)
    fd, w4d, b4d = dev(feat), dev(w4), dev(np.zeros(4, np.float32))
    _lib.check(lib.sta_debug_head_final(h, fd.data_ptr(), w4d.data_ptr(), b4d.data_ptr(),
                                        npix, pts.data_ptr(), conf.data_ptr(), st()))
    res["post_pts"] = max_rel(pts.cpu().numpy().reshape(pin.shape[2], pin.shape[3], 3), g["post_pts"][0])
    res["post_conf"] = max_rel(conf.cpu().numpy().reshape(pin.shape[2], pin.shape[3]), g["post_conf"][0])
    torch.cuda.synchronize()
    return res


# ------------------------------------------------------------------------------------------ end to end
def run_golden_case(name, precision, taps=True, variant=0, frontend=None):
    """HIP forward on the procedural inputs of a golden case; returns {key: rel-L2 error}.  frontend: a frontend that already holds
    the weights the fixture was generated with (the real-checkpoint kit: weights from a FILE) instead of the procedural ones."""
    g, meta = load_golden(name)
    cfg_name = "tiny" if int(meta["cfg_enc_embed_dim"]) == W.TINY.enc_embed_dim else "full"
    if frontend is not None:
        m = frontend
        m.set_precision(precision)
    else:
        m = model(cfg_name, float(meta["qk_gain"]), precision, int(meta["seed"]), int(meta.get("outlier", 0)), hooks=variant != 0)
    set_variant(m, variant)
    cfg = m.cfg
    m.range_report(reset=True)
    H, W_, B, sub = int(meta["H"]), int(meta["W"]), int(meta["B"]), int(meta["sub"])
    gen = W.smooth_images if int(meta["smooth"]) else W.synth_images
    imgs = gen(2 * B, H, W_, seed=int(meta["seed"]), tag=0)
    a, b = dev(imgs[:B]), dev(imgs[B:])
    res = {}
    main, supp = m.forward_pair(a, b)
    torch.cuda.synchronize()
    for side, o in (("main", main), ("supp", supp)):
        pts = o["pts3d_pred"].cpu().numpy(); conf = o["conf"].cpu().numpy()
        res[f"{side}_pts3d"] = rel_l2(pts[:, ::sub, ::sub], g[f"{side}_pts3d"])
        res[f"{side}_conf"] = rel_l2(conf[:, ::sub, ::sub], g[f"{side}_conf"])
        res[f"{side}_pose"] = rel_l2(o["relative_pose"].cpu().numpy(), g[f"{side}_pose"])
        res[f"{side}_pose_conf"] = rel_l2(o["relative_pose_conf"].cpu().numpy(), g[f"{side}_pose_conf"])
        # the same outputs in the max norm (max-abs error / max-abs value): an isolated bad pixel that rel-L2 averages away shows here
        res[f"{side}_pts3d_maxrel"] = max_rel(pts[:, ::sub, ::sub], g[f"{side}_pts3d"])
        res[f"{side}_conf_maxrel"] = max_rel(conf[:, ::sub, ::sub], g[f"{side}_conf"])
        res[f"{side}_pose_maxrel"] = max_rel(o["relative_pose"].cpu().numpy(), g[f"{side}_pose"])
        if "rand_idx" in g:      # off-lattice pixels of the sub-sampled fixtures: every pixel phase of the patch / conv tile / ConvT / bilinear grids
            ri = g["rand_idx"]
            pr, cr = pts.reshape(pts.shape[0], -1, 3)[:, ri], conf.reshape(conf.shape[0], -1)[:, ri]
            res[f"{side}_pts3d_rand"] = rel_l2(pr, g[f"{side}_pts3d_rand"]); res[f"{side}_pts3d_rand_maxrel"] = max_rel(pr, g[f"{side}_pts3d_rand"])
            res[f"{side}_conf_rand"] = rel_l2(cr, g[f"{side}_conf_rand"]); res[f"{side}_conf_rand_maxrel"] = max_rel(cr, g[f"{side}_conf_rand"])
        res[f"{side}_pts3d_norm"] = abs(float(np.sqrt((pts.astype(np.float64) ** 2).sum(axis=(1, 2, 3)))[0]) / float(g[f"{side}_pts3d_l2"][0]) - 1.0)
    # split entry points (what slam.py calls): encoder features + decoder hooks
    ts = torch.tensor([[H, W_]] * B)
    fa, pa = m._encode_image(a, ts, normalize=False)
    fb, pb = m._encode_image(b, ts, normalize=False)
    d1, d2 = m._decode_stereo(fa, fb, pa, pb)
    torch.cuda.synchronize()
    tsub = max(1, sub)
    # the integer output of _encode_image (PositionGetter, sta_blocks.py:241-247; slam.py:144 stores it per keyframe and feeds it
    # back at :162): bit-exact - dtype, shape and every entry (0.0 = identical, 1.0 = not; the callers compare with a tolerance)
    for key, got in (("pos_a", pa), ("pos_b", pb)):
        want = g[key]
        gotn = got.cpu().numpy()
        res[key] = 0.0 if (got.dtype == torch.int64 and gotn.shape == want.shape and want.dtype == np.int64 and np.array_equal(gotn, want)) else 1.0
    res["enc_feat_a"] = rel_l2(fa.cpu().numpy()[:, ::tsub], g["enc_feat_a"])
    res["enc_feat_b"] = rel_l2(fb.cpu().numpy()[:, ::tsub], g["enc_feat_b"])
    for hk in cfg.hooks[1:]:
        res[f"dec1_hook{hk - 1}"] = rel_l2(d1[hk - 1].cpu().numpy()[:, ::tsub], g[f"dec1_hook{hk - 1}"])
        res[f"dec2_hook{hk - 1}"] = rel_l2(d2[hk - 1].cpu().numpy()[:, ::tsub], g[f"dec2_hook{hk - 1}"])
    # split heads == monolithic forward
    pose = m.head_pose_s(d1[-1][:, 0, :])
    hp = m.head_pts([fa] + [t[:, 1:, :] for t in d1], ts)
    torch.cuda.synchronize()
    res["split_pose_vs_golden"] = rel_l2(pose["pose"].cpu().numpy(), g["main_pose"])
    res["split_pts_vs_golden"] = rel_l2(hp["pts3d"].cpu().numpy()[:, ::sub, ::sub], g["main_pts3d"])
    if taps and "dec1_in" in g:
        res["dec1_in"] = rel_l2(d1[0].cpu().numpy(), g["dec1_in"])
    global last_range
    last_range = m.range_report(reset=True)
    return res
