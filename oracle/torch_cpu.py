"""torch-CPU (MKL / oneDNN) re-expression of the reference STA forward.  TEST INFRASTRUCTURE ONLY.

Purpose: the CPU baseline of bench.py on the GPU box's host cores.  The reference's own PyTorch path cannot travel
(its source stays in the build container), and the C + OpenMP oracle (sta_oracle_ops.c) is a clarity-first
restatement that reaches < 100 GFLOP/s - slower than the reference's own CPU figure - so it flatters the GPU.  This
file re-expresses the same algorithm with the torch CPU kernels the reference itself would run on (F.linear,
F.layer_norm, F.gelu, F.conv2d, F.conv_transpose2d, F.interpolate, softmax attention), written from the same
file:line citations as oracle/sta_oracle.py.  It is pinned against the reference goldens by
tests/test_oracle_golden.py::test_torch_cpu_port_vs_reference_golden and never imported by the product path.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F


def _t(sd):
    return {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in sd.items()}


def rope2d(tok, pos, base=100.0):
    """RoPE2D.forward (pos_embed/pos_embed.py:169-185): tok (B,H,N,D), pos (B,N,2) int64 (y,x); D/2 per axis."""
    D = tok.shape[-1] // 2
    inv = 1.0 / (base ** (torch.arange(0, D, 2, dtype=torch.float32) / D))
    out = []
    for ax in range(2):
        t = tok[..., ax * D:(ax + 1) * D]
        ang = pos[:, None, :, ax, None].float() * inv                        # (B,1,N,D/2)
        cos, sin = torch.cat([ang.cos(), ang.cos()], -1), torch.cat([ang.sin(), ang.sin()], -1)
        rot = torch.cat([-t[..., D // 2:], t[..., :D // 2]], -1)
        out.append(t * cos + rot * sin)
    return torch.cat(out, -1)


def _attn(q, k, v, heads, qpos, kpos, base):
    """softmax(rope(q) rope(k)^T / sqrt(64)) v (sta_blocks.py:129-148,188-208); q (B,Nq,C), k/v (B,Nk,C)."""
    B, Nq, Cd = q.shape
    Nk = k.shape[1]
    hd = Cd // heads
    q = rope2d(q.view(B, Nq, heads, hd).transpose(1, 2), qpos, base)
    k = rope2d(k.view(B, Nk, heads, hd).transpose(1, 2), kpos, base)
    v = v.view(B, Nk, heads, hd).transpose(1, 2)
    a = torch.softmax((q @ k.transpose(-1, -2)) * (hd ** -0.5), dim=-1)
    return (a @ v).transpose(1, 2).reshape(B, Nq, Cd)


def _ln(sd, p, x, eps):
    return F.layer_norm(x, (x.shape[-1],), sd[p + ".weight"], sd[p + ".bias"], eps)


def _mlp(sd, p, x):
    return F.linear(F.gelu(F.linear(x, sd[p + ".fc1.weight"], sd[p + ".fc1.bias"])), sd[p + ".fc2.weight"], sd[p + ".fc2.bias"])


def _block(sd, p, x, pos, heads, cfg):
    """Block.forward (sta_blocks.py:166-169)."""
    h = _ln(sd, p + ".norm1", x, cfg.ln_eps)
    q, k, v = F.linear(h, sd[p + ".attn.qkv.weight"], sd[p + ".attn.qkv.bias"]).chunk(3, -1)
    x = x + F.linear(_attn(q, k, v, heads, pos, pos, cfg.rope_base), sd[p + ".attn.proj.weight"], sd[p + ".attn.proj.bias"])
    return x + _mlp(sd, p + ".mlp", _ln(sd, p + ".norm2", x, cfg.ln_eps))


def _dec_block(sd, p, x, y, xpos, ypos, heads, cfg):
    """DecoderBlock.forward (sta_blocks.py:226-231)."""
    h = _ln(sd, p + ".norm1", x, cfg.ln_eps)
    q, k, v = F.linear(h, sd[p + ".attn.qkv.weight"], sd[p + ".attn.qkv.bias"]).chunk(3, -1)
    x = x + F.linear(_attn(q, k, v, heads, xpos, xpos, cfg.rope_base), sd[p + ".attn.proj.weight"], sd[p + ".attn.proj.bias"])
    yn = _ln(sd, p + ".norm_y", y, cfg.ln_eps)
    c = p + ".cross_attn"
    q = F.linear(_ln(sd, p + ".norm2", x, cfg.ln_eps), sd[c + ".projq.weight"], sd[c + ".projq.bias"])
    k = F.linear(yn, sd[c + ".projk.weight"], sd[c + ".projk.bias"])
    v = F.linear(yn, sd[c + ".projv.weight"], sd[c + ".projv.bias"])
    x = x + F.linear(_attn(q, k, v, heads, xpos, ypos, cfg.rope_base), sd[c + ".proj.weight"], sd[c + ".proj.bias"])
    return x + _mlp(sd, p + ".mlp", _ln(sd, p + ".norm3", x, cfg.ln_eps))


def _positions(B, hp, wp):
    y, x = torch.meshgrid(torch.arange(hp), torch.arange(wp), indexing="ij")
    return torch.stack([y.reshape(-1), x.reshape(-1)], -1)[None].expand(B, -1, -1)


def encode(cfg, sd, img):
    """_encode_image(normalize=False) (sta_model.py:163-174)."""
    B, _c, H, W_ = img.shape
    x = F.conv2d(img, sd["patch_embed.proj.weight"], sd["patch_embed.proj.bias"], stride=16).flatten(2).transpose(1, 2)
    pos = _positions(B, H // 16, W_ // 16)
    for i in range(cfg.enc_depth):
        x = _block(sd, f"enc_blocks.{i}", x, pos, cfg.enc_num_heads, cfg)
    return x, pos


def decode(cfg, sd, f1, f2, p1, p2):
    """_decode_stereo (sta_model.py:177-244)."""
    B = f1.shape[0]
    tok = sd["init_pose_token"].expand(B, -1, -1)
    a = torch.cat([tok, F.linear(f1, sd["decoder_embed.weight"], sd["decoder_embed.bias"])], 1)
    b = torch.cat([tok, F.linear(f2, sd["decoder_embed.weight"], sd["decoder_embed.bias"])], 1)
    m1 = -torch.ones(B, 1, 2, dtype=p1.dtype)
    p1, p2 = torch.cat([m1, p1], 1), torch.cat([m1, p2], 1)
    l1, l2 = [a], [b]
    for i in range(cfg.dec_depth):
        a, b = l1[-1], l2[-1]
        l1.append(_dec_block(sd, f"dec_block.{i}", a, b, p1, p2, cfg.dec_num_heads, cfg))
        l2.append(_dec_block(sd, f"dec_block.{i}", b, a, p2, p1, cfg.dec_num_heads, cfg))
    l1[-1] = _ln(sd, "dec_norm", l1[-1], cfg.ln_eps)
    l2[-1] = _ln(sd, "dec_norm", l2[-1], cfg.ln_eps)
    return l1, l2


def head_pose(cfg, sd, tok):
    """PoseHead_small.forward (heads/pose_head.py:38-57,94-120)."""
    h = tok
    for i in (0, 2, 4):
        h = F.relu(F.linear(h, sd[f"head_pose_s.mlp.{i}.weight"], sd[f"head_pose_s.mlp.{i}.bias"]))
    t = F.linear(h, sd["head_pose_s.fc_t.weight"], sd["head_pose_s.fc_t.bias"])
    r = F.linear(h, sd["head_pose_s.fc_rot.weight"], sd["head_pose_s.fc_rot.bias"]).view(-1, 3, 3)
    c = torch.sigmoid(F.linear(h, sd["head_pose_s.fc_conf.0.weight"], sd["head_pose_s.fc_conf.0.bias"]))[:, 0]
    m = F.normalize(r, p=2, dim=-1).transpose(1, 2)
    u, _s, v = torch.svd(m)
    det = torch.det(v @ u.transpose(1, 2))
    R = torch.cat([v[:, :, :2], v[:, :, 2:] * det.view(-1, 1, 1)], 2) @ u.transpose(1, 2)
    pose = torch.eye(4).repeat(tok.shape[0], 1, 1)
    pose[:, :3, :3] = R
    pose[:, :3, 3] = t
    return pose, c


def _rcu(sd, p, x):
    o = F.conv2d(F.relu(x), sd[p + ".conv1.weight"], sd[p + ".conv1.bias"], padding=1)
    return F.conv2d(F.relu(o), sd[p + ".conv2.weight"], sd[p + ".conv2.bias"], padding=1) + x


def _fusion(sd, p, x0, x1=None):
    o = x0 if x1 is None else x0 + _rcu(sd, p + ".resConfUnit1", x1)
    o = F.interpolate(_rcu(sd, p + ".resConfUnit2", o), scale_factor=2, mode="bilinear", align_corners=True)
    return F.conv2d(o, sd[p + ".out_conv.weight"], sd[p + ".out_conv.bias"])


def head_pts(cfg, sd, tokens, H, W_):
    """DPTOutputAdapter_fix.forward + postprocess (heads/dpt_head.py:34-66, dpt_block.py, postprocess.py:10-62)."""
    dp = "downstream_head_pts.dpt."
    hp, wp = H // 16, W_ // 16
    L = [tokens[hk].transpose(1, 2).reshape(tokens[hk].shape[0], -1, hp, wp) for hk in cfg.hooks]
    a = dp + "act_postprocess."
    l0 = F.conv_transpose2d(F.conv2d(L[0], sd[a + "0.0.weight"], sd[a + "0.0.bias"]), sd[a + "0.1.weight"], sd[a + "0.1.bias"], stride=4)
    l1 = F.conv_transpose2d(F.conv2d(L[1], sd[a + "1.0.weight"], sd[a + "1.0.bias"]), sd[a + "1.1.weight"], sd[a + "1.1.bias"], stride=2)
    l2 = F.conv2d(L[2], sd[a + "2.0.weight"], sd[a + "2.0.bias"])
    l3 = F.conv2d(F.conv2d(L[3], sd[a + "3.0.weight"], sd[a + "3.0.bias"]), sd[a + "3.1.weight"], sd[a + "3.1.bias"], stride=2, padding=1)
    ls = [F.conv2d(l, sd[f"{dp}scratch.layer_rn.{i}.weight"], None, padding=1) for i, l in enumerate((l0, l1, l2, l3))]
    s = dp + "scratch.refinenet"
    path = _fusion(sd, s + "4", ls[3])[:, :, :ls[2].shape[2], :ls[2].shape[3]]
    for r, l in ((3, ls[2]), (2, ls[1]), (1, ls[0])):
        path = _fusion(sd, s + str(r), path, l)
    o = F.conv2d(path, sd[dp + "head.0.weight"], sd[dp + "head.0.bias"], padding=1)
    o = F.interpolate(o, scale_factor=2, mode="bilinear", align_corners=True)
    o = F.conv2d(F.relu(F.conv2d(o, sd[dp + "head.2.weight"], sd[dp + "head.2.bias"], padding=1)), sd[dp + "head.4.weight"], sd[dp + "head.4.bias"])
    fmap = o.permute(0, 2, 3, 1)
    xyz = fmap[..., :3]
    d = xyz.norm(dim=-1, keepdim=True)
    pts = xyz / d.clip(min=1e-8) * torch.expm1(d)
    conf = 1 + fmap[..., 3].exp()
    if H > W_:      # portrait: transposed(head(decout, (H, W))) (utils/misc.py:60-61,81)
        pts, conf = pts.swapaxes(1, 2), conf.swapaxes(1, 2)
    return pts, conf


@torch.no_grad()
def forward_pair(cfg, sd_np, img_a, img_b):
    """SymmetricTwoViewAssociation.forward with one neighbour view (sta_model.py:247-291) -> numpy outputs."""
    sd = _t(sd_np)
    a, b = torch.from_numpy(np.ascontiguousarray(img_a)), torch.from_numpy(np.ascontiguousarray(img_b))
    H, W_ = a.shape[2], a.shape[3]
    fa, pa = encode(cfg, sd, a)
    fb, pb = encode(cfg, sd, b)
    d1, d2 = decode(cfg, sd, fa, fb, pa, pb)
    res = {}
    for key, enc, dec in (("main", fa, d1), ("supp", fb, d2)):
        pts, conf = head_pts(cfg, sd, [enc] + [t[:, 1:, :] for t in dec], H, W_)
        pose, pconf = head_pose(cfg, sd, dec[-1][:, 0, :])
        res[key] = {"pts3d": pts.numpy(), "conf": conf.numpy(), "pose": pose.numpy(), "pose_conf": pconf.numpy()}
    return res
