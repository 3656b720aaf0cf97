"""CPU oracle: restatement of the reference STA forward (fp32), composed from the C ops of
sta_oracle_ops.c.  TEST INFRASTRUCTURE ONLY - imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py as the checker; the product path never touches it.

Parity of THIS file is pinned by tests/test_oracle_golden.py against golden vectors generated from
the reference PyTorch model itself (tests/golden/*.npz <- oracle/gen_golden.py).

Every function mirrors one reference function (paths relative to vista_slam/sta_model/).
Weights: dict name -> float32 numpy array with the reference state_dict keys.
"""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
_lib = None
_f = C.POINTER(C.c_float)


def lib():
    global _lib
    if _lib is None:
        path = os.path.join(HERE, "libsta_oracle.so")
        if not os.path.exists(path):
            from . import build_oracle
            build_oracle.build()
        _lib = C.CDLL(path)
    return _lib


def _p(a):
    return a.ctypes.data_as(_f)


def _c(a):
    return np.ascontiguousarray(a, dtype=np.float32)


# ------------------------------------------------------------------ ops
def linear(x, w, b):
    x = _c(x); sh = x.shape
    x2 = x.reshape(-1, sh[-1])
    y = np.empty((x2.shape[0], w.shape[0]), np.float32)
    lib().o_linear(_p(x2), _p(_c(w)), _p(_c(b)) if b is not None else None, _p(y), x2.shape[0], w.shape[0], sh[-1])
    return y.reshape(sh[:-1] + (w.shape[0],))


def layernorm(x, g, b, eps=1e-6):
    x = _c(x); y = np.empty_like(x)
    lib().o_layernorm(_p(x), _p(_c(g)), _p(_c(b)), _p(y), x.size // x.shape[-1], x.shape[-1], C.c_float(eps))
    return y


def gelu(x):
    x = _c(x).copy()
    lib().o_gelu(_p(x), C.c_int64(x.size))
    return x


def rope2d(tok, pos, base=100.0):
    """tok (B,H,N,D) -> rotated copy; pos (B,N,2) int64  (pos_embed/pos_embed.py:169-185)."""
    tok = _c(tok).copy(); pos = np.ascontiguousarray(pos, dtype=np.int64)
    B, H, N, D = tok.shape
    lib().o_rope2d(_p(tok), pos.ctypes.data_as(C.POINTER(C.c_int64)), B, H, N, D, C.c_float(base))
    return tok


def attention(q, k, v, scale):
    q, k, v = _c(q), _c(k), _c(v)
    B, H, Nq, D = q.shape
    out = np.empty((B, Nq, H * D), np.float32)
    lib().o_attention(_p(q), _p(k), _p(v), _p(out), B, H, Nq, k.shape[2], D, C.c_float(scale))
    return out


def conv2d(x, w, b, stride=1, pad=0):
    x, w = _c(x), _c(w)
    B, Ci, H, W_ = x.shape
    Co, _, k, _ = w.shape
    Ho, Wo = (H + 2 * pad - k) // stride + 1, (W_ + 2 * pad - k) // stride + 1
    y = np.empty((B, Co, Ho, Wo), np.float32)
    lib().o_conv2d(_p(x), _p(w), _p(_c(b)) if b is not None else None, _p(y), B, Ci, H, W_, Co, k, stride, pad)
    return y


def conv_transpose2d(x, w, b, k):
    x, w = _c(x), _c(w)
    B, Ci, H, W_ = x.shape
    Co = w.shape[1]
    y = np.empty((B, Co, H * k, W_ * k), np.float32)
    lib().o_conv_transpose2d(_p(x), _p(w), _p(_c(b)), _p(y), B, Ci, H, W_, Co, k)
    return y


def bilinear_up2(x):
    x = _c(x)
    B, Cc, H, W_ = x.shape
    y = np.empty((B, Cc, 2 * H, 2 * W_), np.float32)
    lib().o_bilinear_up2(_p(x), _p(y), B * Cc, H, W_)
    return y


def svd_orthogonalize(m):
    m = _c(m).reshape(-1, 3, 3)
    r = np.empty_like(m)
    lib().o_svd_orthogonalize(_p(m), _p(r), m.shape[0])
    return r


def postprocess(out):
    out = _c(out)
    B, _, H, W_ = out.shape
    pts = np.empty((B, H, W_, 3), np.float32); conf = np.empty((B, H, W_), np.float32)
    lib().o_postprocess(_p(out), _p(pts), _p(conf), B, H, W_)
    return pts, conf


# ------------------------------------------------------------------ model (mirrors the reference)
def positions(B, hp, wp):
    """PositionGetter (blocks/sta_blocks.py:241-247): cartesian_prod(arange(h), arange(w)) -> (y,x)."""
    yy, xx = np.meshgrid(np.arange(hp), np.arange(wp), indexing="ij")
    p = np.stack([yy.ravel(), xx.ravel()], -1).astype(np.int64)
    return np.broadcast_to(p[None], (B,) + p.shape).copy()


def patch_embed(sd, img):
    """PatchEmbedDust3R.forward (patch_embed.py:17-27): conv k=s=16, flatten(2).transpose(1,2)."""
    x = conv2d(img, sd["patch_embed.proj.weight"], sd["patch_embed.proj.bias"], stride=16, pad=0)
    B, E, hp, wp = x.shape
    return x.reshape(B, E, hp * wp).transpose(0, 2, 1).copy(), positions(B, hp, wp)


def self_attention(sd, p, x, pos, heads, base):
    """XFormer_Attention.forward (blocks/sta_blocks.py:129-148)."""
    B, N, Cd = x.shape
    qkv = linear(x, sd[p + ".qkv.weight"], sd[p + ".qkv.bias"]).reshape(B, N, 3, heads, Cd // heads).transpose(2, 0, 3, 1, 4)
    q, k, v = rope2d(qkv[0], pos, base), rope2d(qkv[1], pos, base), qkv[2]
    o = attention(q, k, v, (Cd // heads) ** -0.5)
    return linear(o, sd[p + ".proj.weight"], sd[p + ".proj.bias"])


def mlp(sd, p, x):
    """Mlp.forward (blocks/sta_blocks.py:73-79)."""
    return linear(gelu(linear(x, sd[p + ".fc1.weight"], sd[p + ".fc1.bias"])), sd[p + ".fc2.weight"], sd[p + ".fc2.bias"])


def ln(sd, p, x, eps):
    return layernorm(x, sd[p + ".weight"], sd[p + ".bias"], eps)


def block(sd, p, x, pos, heads, cfg):
    """Block.forward (blocks/sta_blocks.py:166-169)."""
    x = x + self_attention(sd, p + ".attn", ln(sd, p + ".norm1", x, cfg.ln_eps), pos, heads, cfg.rope_base)
    return x + mlp(sd, p + ".mlp", ln(sd, p + ".norm2", x, cfg.ln_eps))


def cross_attention(sd, p, xq, y, qpos, kpos, heads, base):
    """CrossAttention.forward (blocks/sta_blocks.py:188-208)."""
    B, Nq, Cd = xq.shape
    Nk = y.shape[1]
    hd = Cd // heads
    q = linear(xq, sd[p + ".projq.weight"], sd[p + ".projq.bias"]).reshape(B, Nq, heads, hd).transpose(0, 2, 1, 3)
    k = linear(y, sd[p + ".projk.weight"], sd[p + ".projk.bias"]).reshape(B, Nk, heads, hd).transpose(0, 2, 1, 3)
    v = linear(y, sd[p + ".projv.weight"], sd[p + ".projv.bias"]).reshape(B, Nk, heads, hd).transpose(0, 2, 1, 3)
    o = attention(rope2d(q, qpos, base), rope2d(k, kpos, base), v, hd ** -0.5)
    return linear(o, sd[p + ".proj.weight"], sd[p + ".proj.bias"])


def decoder_block(sd, p, x, y, xpos, ypos, heads, cfg):
    """DecoderBlock.forward (blocks/sta_blocks.py:226-231)."""
    x = x + self_attention(sd, p + ".attn", ln(sd, p + ".norm1", x, cfg.ln_eps), xpos, heads, cfg.rope_base)
    y_ = ln(sd, p + ".norm_y", y, cfg.ln_eps)
    x = x + cross_attention(sd, p + ".cross_attn", ln(sd, p + ".norm2", x, cfg.ln_eps), y_, xpos, ypos, heads, cfg.rope_base)
    return x + mlp(sd, p + ".mlp", ln(sd, p + ".norm3", x, cfg.ln_eps))


def encode_image(cfg, sd, img):
    """_encode_image(normalize=False) (sta_model.py:163-174)."""
    x, pos = patch_embed(sd, img)
    for i in range(cfg.enc_depth):
        x = block(sd, f"enc_blocks.{i}", x, pos, cfg.enc_num_heads, cfg)
    return x, pos


def decode_stereo(cfg, sd, feat1, feat2, pos1, pos2):
    """_decode_stereo (sta_model.py:177-244)."""
    B = feat1.shape[0]
    tok = np.broadcast_to(sd["init_pose_token"], (B, 1, cfg.dec_embed_dim))
    f1 = np.concatenate([tok, linear(feat1, sd["decoder_embed.weight"], sd["decoder_embed.bias"])], 1)
    f2 = np.concatenate([tok, linear(feat2, sd["decoder_embed.weight"], sd["decoder_embed.bias"])], 1)
    m1 = -np.ones((B, 1, 2), np.int64)
    p1 = np.concatenate([m1, pos1], 1); p2 = np.concatenate([m1, pos2], 1)
    final1, final2 = [f1], [f2]
    for i in range(cfg.dec_depth):
        a, b = final1[-1], final2[-1]
        o1 = decoder_block(sd, f"dec_block.{i}", a, b, p1, p2, cfg.dec_num_heads, cfg)
        o2 = decoder_block(sd, f"dec_block.{i}", b, a, p2, p1, cfg.dec_num_heads, cfg)
        final1.append(o1); final2.append(o2)
    final1[-1] = ln(sd, "dec_norm", final1[-1], cfg.ln_eps)
    final2[-1] = ln(sd, "dec_norm", final2[-1], cfg.ln_eps)
    return final1, final2


def head_pose(cfg, sd, tok):
    """PoseHead_small.forward (heads/pose_head.py:109-120)."""
    h = tok
    for i in (0, 2, 4):
        h = np.maximum(linear(h, sd[f"head_pose_s.mlp.{i}.weight"], sd[f"head_pose_s.mlp.{i}.bias"]), 0)
    t = linear(h, sd["head_pose_s.fc_t.weight"], sd["head_pose_s.fc_t.bias"])
    r = linear(h, sd["head_pose_s.fc_rot.weight"], sd["head_pose_s.fc_rot.bias"])
    c = linear(h, sd["head_pose_s.fc_conf.0.weight"], sd["head_pose_s.fc_conf.0.bias"])[:, 0]
    B = tok.shape[0]
    pose = np.zeros((B, 4, 4), np.float32)
    pose[:, :3, :3] = svd_orthogonalize(r)
    pose[:, :3, 3] = t
    pose[:, 3, 3] = 1.0
    return pose, (1.0 / (1.0 + np.exp(-c.astype(np.float32)))).astype(np.float32)


def _rcu(sd, p, x):
    """ResidualConvUnit_custom.forward (heads/dpt_block.py:121-142): input ReLU is not in-place."""
    out = conv2d(np.maximum(x, 0), sd[p + ".conv1.weight"], sd[p + ".conv1.bias"], 1, 1)
    out = conv2d(np.maximum(out, 0), sd[p + ".conv2.weight"], sd[p + ".conv2.bias"], 1, 1)
    return out + x


def _fusion(sd, p, x0, x1=None):
    """FeatureFusionBlock_custom.forward (heads/dpt_block.py:189-218), width_ratio == 1."""
    out = x0
    if x1 is not None:
        out = out + _rcu(sd, p + ".resConfUnit1", x1)
    out = _rcu(sd, p + ".resConfUnit2", out)
    out = bilinear_up2(out)
    return conv2d(out, sd[p + ".out_conv.weight"], sd[p + ".out_conv.bias"], 1, 0)


def head_pts(cfg, sd, tokens, H, W_):
    """DPTOutputAdapter_fix.forward + postprocess (heads/dpt_head.py:34-66, heads/postprocess.py:10-62).
    tokens: list of dec_depth+2 arrays ([enc_feat] + decoder outputs without the pose token)."""
    dp = "downstream_head_pts.dpt."
    hp, wp = H // 16, W_ // 16
    layers = []
    for i, hk in enumerate(cfg.hooks):
        t = tokens[hk]
        B, N, Cd = t.shape
        layers.append(t.transpose(0, 2, 1).reshape(B, Cd, hp, wp))           # 'b (nh nw) c -> b c nh nw'
    a = f"{dp}act_postprocess."
    l0 = conv_transpose2d(conv2d(layers[0], sd[a + "0.0.weight"], sd[a + "0.0.bias"]), sd[a + "0.1.weight"], sd[a + "0.1.bias"], 4)
    l1 = conv_transpose2d(conv2d(layers[1], sd[a + "1.0.weight"], sd[a + "1.0.bias"]), sd[a + "1.1.weight"], sd[a + "1.1.bias"], 2)
    l2 = conv2d(layers[2], sd[a + "2.0.weight"], sd[a + "2.0.bias"])
    l3 = conv2d(conv2d(layers[3], sd[a + "3.0.weight"], sd[a + "3.0.bias"]), sd[a + "3.1.weight"], sd[a + "3.1.bias"], 2, 1)
    ls = [conv2d(l, sd[f"{dp}scratch.layer_rn.{i}.weight"], None, 1, 1) for i, l in enumerate((l0, l1, l2, l3))]
    s = dp + "scratch.refinenet"
    path4 = _fusion(sd, s + "4", ls[3])[:, :, :ls[2].shape[2], :ls[2].shape[3]]
    path3 = _fusion(sd, s + "3", path4, ls[2])
    path2 = _fusion(sd, s + "2", path3, ls[1])
    path1 = _fusion(sd, s + "1", path2, ls[0])
    o = conv2d(path1, sd[dp + "head.0.weight"], sd[dp + "head.0.bias"], 1, 1)
    o = bilinear_up2(o)
    o = np.maximum(conv2d(o, sd[dp + "head.2.weight"], sd[dp + "head.2.bias"], 1, 1), 0)
    o = conv2d(o, sd[dp + "head.4.weight"], sd[dp + "head.4.bias"])
    pts, conf = postprocess(o)
    if H > W_:      # portrait: model.head_pts = transpose_to_landscape(head) returns transposed(head(decout, (H, W))) (utils/misc.py:60-61,81)
        pts, conf = pts.swapaxes(1, 2), conf.swapaxes(1, 2)
    return pts, conf


def forward_pair(cfg, sd, img_a, img_b):
    """SymmetricTwoViewAssociation.forward with one neighbour view (sta_model.py:247-291)."""
    H, W_ = img_a.shape[2], img_a.shape[3]
    fa, pa = encode_image(cfg, sd, img_a)
    fb, pb = encode_image(cfg, sd, img_b)
    d1, d2 = decode_stereo(cfg, sd, fa, fb, pa, pb)
    res = {"enc_feat_a": fa, "enc_feat_b": fb, "dec1": d1, "dec2": d2}
    for key, enc, dec in (("main", fa, d1), ("supp", fb, d2)):
        pts, conf = head_pts(cfg, sd, [enc] + [t[:, 1:, :] for t in dec], H, W_)
        pose, pconf = head_pose(cfg, sd, dec[-1][:, 0, :])
        res[key] = {"pts3d": pts, "conf": conf, "pose": pose, "pose_conf": pconf}
    return res


def encode_image_normalized(cfg, sd, img):
    """_encode_image(normalize=True) (sta_model.py:163-174): the blocks, then enc_norm."""
    x, pos = encode_image(cfg, sd, img)
    return ln(sd, "enc_norm", x, cfg.ln_eps), pos


# ------------------------------------------------------------------ keyframe edge regression
def regress_two_views(cfg, sd, enc_feat_i, enc_feat_j, pos_i, pos_j, adjacent, rel_pose_thres, H, W_):
    """OnlineSLAM.regress_two_views (vista_slam/slam.py:153-189) for one edge (i, j), B = 1:
    _decode_stereo -> head_pose_s on the ij pose token -> early return when `conf < rel_pose_thres and i-j != 1`
    (:169-170) -> head_pts on both sides (ji first, :179-180) -> pcls = cat(ij, ji), confs = cat(ij, ji) ->
    estimate_intrinsic_from_pts3d(shared_intrinsic=True) -> depths = pcls[..., 2].
    Returns (pose_ij 4x4, rel_pose_conf_ij, confs, intri, depths); the last three are None for a rejected edge.
    (The reference converts pose_ij with pp.mat2SE3 - see mat_to_se3 below; pypose is absent here.)"""
    dec_ij, dec_ji = decode_stereo(cfg, sd, enc_feat_i, enc_feat_j, pos_i, pos_j)
    pose, conf = head_pose(cfg, sd, dec_ij[-1][:, 0, :])
    if float(conf[0]) < rel_pose_thres and not adjacent:
        return pose[0], float(conf[0]), None, None, None
    pts_ji, conf_ji = head_pts(cfg, sd, [enc_feat_j] + [t[:, 1:, :] for t in dec_ji], H, W_)
    pts_ij, conf_ij = head_pts(cfg, sd, [enc_feat_i] + [t[:, 1:, :] for t in dec_ij], H, W_)
    pcls = np.concatenate([pts_ij, pts_ji], 0)
    confs = np.concatenate([conf_ij, conf_ji], 0)
    intri = estimate_intrinsic_from_pts3d(pcls, confs, shared_intrinsic=True)
    return pose[0], float(conf[0]), confs, intri, pcls[..., 2]


# ------------------------------------------------------------------ post-STA reductions
def estimate_intrinsic_from_pts3d(pts3d, confidence, shared_intrinsic=False):
    """vista_slam/utils/slam_utils.py:8-79, restated in numpy float32 (same formula, same clamps)."""
    pts3d = np.asarray(pts3d, np.float32); confidence = np.asarray(confidence, np.float32)
    B, H, W_, _ = pts3d.shape
    cx, cy = np.float32(W_ / 2.0), np.float32(H / 2.0)
    v, u = np.meshgrid(np.arange(H), np.arange(W_), indexing="ij")
    u = (u.astype(np.float32) - cx).reshape(1, -1)
    v = (v.astype(np.float32) - cy).reshape(1, -1)
    X, Y, Z = (pts3d[..., k].reshape(B, -1) for k in range(3))
    w = np.maximum(confidence.reshape(B, -1), np.float32(1e-6))
    with np.errstate(divide="ignore", invalid="ignore"):
        xz = np.nan_to_num(X / Z, nan=0.0, posinf=0.0, neginf=0.0).astype(np.float32)
        yz = np.nan_to_num(Y / Z, nan=0.0, posinf=0.0, neginf=0.0).astype(np.float32)
    if shared_intrinsic:
        fx = (w * xz * u).sum(dtype=np.float64) / (w * xz ** 2).sum(dtype=np.float64)
        fy = (w * yz * v).sum(dtype=np.float64) / (w * yz ** 2).sum(dtype=np.float64)
        return np.array([[fx, 0, cx], [0, fy, cy], [0, 0, 1]], np.float32)
    fx = (w * xz * u).sum(axis=1, dtype=np.float64) / (w * xz ** 2).sum(axis=1, dtype=np.float64)
    fy = (w * yz * v).sum(axis=1, dtype=np.float64) / (w * yz ** 2).sum(axis=1, dtype=np.float64)
    K = np.zeros((B, 3, 3), np.float32)
    K[:, 0, 0], K[:, 1, 1], K[:, 0, 2], K[:, 1, 2], K[:, 2, 2] = fx, fy, cx, cy, 1.0
    return K


def estimate_scale_with_depth_and_confidence(Di, Dj, ci, cj):
    """vista_slam/utils/slam_utils.py:168-190."""
    Di, Dj, ci, cj = (np.asarray(t, np.float32).reshape(-1) for t in (Di, Dj, ci, cj))
    w = np.maximum(ci * cj, np.float32(1e-6))
    return np.float32((w * Di * Dj).sum(dtype=np.float64) / (w * Di * Di).sum(dtype=np.float64))


def world_pointcloud(depths, scales, intrinsics, poses, confs, imgs, thres):
    """slam.py:396-408 + compute_local_pointclouds (slam_utils.py:82-121), numpy fp32: -> (points [M,3], colors [M,3])
    in boolean-mask (view-major, row-major) order."""
    N, H, W = depths.shape
    y, x = np.meshgrid(np.arange(H, dtype=np.float32), np.arange(W, dtype=np.float32), indexing="ij")
    pix = np.stack([x, y, np.ones_like(x)], -1).reshape(-1, 3)                       # (H*W, 3)
    Kinv = np.linalg.inv(intrinsics.astype(np.float32)).astype(np.float32)
    cam = np.einsum("nij,pj->npi", Kinv, pix).astype(np.float32)                     # [N, H*W, 3]
    local = cam * (depths * scales.reshape(N, 1, 1)).reshape(N, -1, 1).astype(np.float32)
    hom = np.concatenate([local, np.ones((N, H * W, 1), np.float32)], -1)
    world = np.einsum("npj,nij->npi", hom, poses.astype(np.float32))[..., :3].reshape(N, H, W, 3)
    mask = confs > thres
    images = (imgs.transpose(0, 2, 3, 1).astype(np.float32) + np.float32(1.0)) / np.float32(2.0)
    return world[mask].astype(np.float32), images[mask]


def mat_to_se3(pose):
    """pp.mat2SE3 (slam.py:166) by definition: (tx,ty,tz,qx,qy,qz,qw), qw >= 0 (Shepperd's method, float64)."""
    pose = np.asarray(pose, np.float64).reshape(-1, 4, 4)
    out = np.zeros((pose.shape[0], 7))
    for b, P in enumerate(pose):
        m = P[:3, :3]
        tr = np.trace(m)
        if tr > 0:
            s = np.sqrt(tr + 1.0) * 2; q = [(m[2, 1] - m[1, 2]) / s, (m[0, 2] - m[2, 0]) / s, (m[1, 0] - m[0, 1]) / s, 0.25 * s]
        elif m[0, 0] > m[1, 1] and m[0, 0] > m[2, 2]:
            s = np.sqrt(1.0 + m[0, 0] - m[1, 1] - m[2, 2]) * 2; q = [0.25 * s, (m[0, 1] + m[1, 0]) / s, (m[0, 2] + m[2, 0]) / s, (m[2, 1] - m[1, 2]) / s]
        elif m[1, 1] > m[2, 2]:
            s = np.sqrt(1.0 + m[1, 1] - m[0, 0] - m[2, 2]) * 2; q = [(m[0, 1] + m[1, 0]) / s, 0.25 * s, (m[1, 2] + m[2, 1]) / s, (m[0, 2] - m[2, 0]) / s]
        else:
            s = np.sqrt(1.0 + m[2, 2] - m[0, 0] - m[1, 1]) * 2; q = [(m[0, 2] + m[2, 0]) / s, (m[1, 2] + m[2, 1]) / s, 0.25 * s, (m[1, 0] - m[0, 1]) / s]
        q = np.array(q); q = q / np.linalg.norm(q) * (1.0 if q[3] >= 0 else -1.0)
        out[b, :3] = P[:3, 3]; out[b, 3:] = q
    return out
