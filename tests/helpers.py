"""Shared test helpers: golden loading, error metrics, plain fp32 reference ops (own restatements,
pinned against tests/golden/ops.npz in test_refops_cpu.py)."""
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")


def load_golden(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    d = {k: z[k] for k in z.files}
    meta = dict(zip([str(k) for k in d.pop("meta_keys")], d.pop("meta_vals").tolist())) if "meta_keys" in d else {}
    return d, meta


def rel_l2(a, b):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    assert a.shape == b.shape, (a.shape, b.shape)
    return float(np.sqrt(((a - b) ** 2).sum()) / max(np.sqrt((b ** 2).sum()), 1e-30))


def max_rel(a, b):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


def rope2d_ref(tok, pos, base=100.0):
    """tok (B,H,N,D) float32, pos (B,N,2) int -> rotated copy (pos_embed.py:169-185 semantics)."""
    tok = np.asarray(tok, np.float32)
    B, H, N, D = tok.shape
    Q = D // 4
    out = tok.copy()
    inv = (1.0 / (np.float32(base) ** (np.arange(Q, dtype=np.float32) / np.float32(Q)))).astype(np.float32)
    for xy in range(2):
        ang = pos[:, None, :, xy, None].astype(np.float32) * inv[None, None, None, :]
        c, s = np.cos(ang).astype(np.float32), np.sin(ang).astype(np.float32)
        u = tok[..., xy * 2 * Q: xy * 2 * Q + Q]
        v = tok[..., xy * 2 * Q + Q: xy * 2 * Q + 2 * Q]
        out[..., xy * 2 * Q: xy * 2 * Q + Q] = u * c - v * s
        out[..., xy * 2 * Q + Q: xy * 2 * Q + 2 * Q] = v * c + u * s
    return out


def grid_pos(B, hp, wp, pose_tok=False):
    yy, xx = np.meshgrid(np.arange(hp), np.arange(wp), indexing="ij")
    p = np.stack([yy.ravel(), xx.ravel()], -1).astype(np.int64)
    if pose_tok:
        p = np.concatenate([np.full((1, 2), -1, np.int64), p], 0)
    return np.broadcast_to(p[None], (B,) + p.shape).copy()
