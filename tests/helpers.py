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


# ---------------------------------------------------------------------------------------------------------
# keyframe-sequence goldens (tests/golden/seq_*.npz, oracle/gen_golden.py gen_seq): one record per candidate edge
def seq_meta(meta):
    return {k: int(meta[k]) for k in ("H", "W", "nkf", "neighbor_edge_num", "loop_edge_num", "loop_dist_min", "sub", "nrand", "seed", "tag")}


def compare_seq_edges(edges, g, meta, tol=1e-3):
    """`edges`: list of dicts {i, j, pose [4,4], conf float, accepted bool, confs [2,H,W] / None, intri, depths, scales [2] with
    NaN for 'no scale edge', scale_confs [2]} (numpy), in the reference's edge order.  Compares every field of every edge of the
    golden `g`: decisions and (i, j) exactly, numbers as rel-L2 AND max-abs / max-abs - both below `tol`.  -> worst errors."""
    m = seq_meta(meta)
    n = int(g["n_edges"])
    assert len(edges) == n, (len(edges), n)
    sub, rand_idx = m["sub"], g.get("rand_idx")
    worst = {}

    def note(key, got, want):
        e1, e2 = rel_l2(got, want), max_rel(got, want)
        worst[key] = max(worst.get(key, 0.0), e1)
        worst[key + "_max"] = max(worst.get(key + "_max", 0.0), e2)
        assert e1 < tol and e2 < tol, (key, e1, e2)

    for e in range(n):
        r = edges[e]
        assert (r["i"], r["j"]) == tuple(int(v) for v in g[f"e{e}_ij"]), (e, r["i"], r["j"], g[f"e{e}_ij"])
        assert bool(r["accepted"]) == bool(g[f"e{e}_accepted"]), (e, r["conf"], float(g["thres"]))          # decisions identical
        note("pose", r["pose"], g[f"e{e}_pose"])
        assert abs(float(r["conf"]) - float(g[f"e{e}_conf"])) < 0.1 * float(g["thres_margin"]), (e, r["conf"], float(g[f"e{e}_conf"]))
        note("pose_conf", np.array([r["conf"]]), np.array([float(g[f"e{e}_conf"])]))
        if not r["accepted"]:
            assert r["confs"] is None and r["intri"] is None and r["depths"] is None
            continue
        note("confs", r["confs"][:, ::sub, ::sub], g[f"e{e}_confs"])
        note("depths", r["depths"][:, ::sub, ::sub], g[f"e{e}_depths"])
        if rand_idx is not None:                                   # off-lattice pixels: every phase of the 16x16 patch / conv tiles
            note("confs_rand", r["confs"].reshape(2, -1)[:, rand_idx], g[f"e{e}_confs_rand"])
            note("depths_rand", r["depths"].reshape(2, -1)[:, rand_idx], g[f"e{e}_depths_rand"])
        note("intri", r["intri"], g[f"e{e}_intri"])
        worst["confs_norm"] = max(worst.get("confs_norm", 0.0), abs(float(np.sqrt((r["confs"].astype(np.float64) ** 2).sum())) / float(g[f"e{e}_confs_l2"]) - 1.0))
        worst["depths_norm"] = max(worst.get("depths_norm", 0.0), abs(float(np.sqrt((r["depths"].astype(np.float64) ** 2).sum())) / float(g[f"e{e}_depths_l2"]) - 1.0))
        for k in range(2):
            want = float(g[f"e{e}_scale"][k])
            got = r["scales"][k]
            assert np.isnan(want) == (got is None or np.isnan(got)), (e, k, want, got)        # the same views get scale edges
            if not np.isnan(want):
                # s = sum(w Di Dj) / sum(w Di Di): with procedural weights the depths have both signs and the numerator cancels
                # (|s| down to 0.06), so the error is judged against the same ratio with |Di Dj| (`scale_abs`, stored by the
                # generator from the reference's tensors) - the magnitude the rounding errors of the maps actually scale with
                sabs = float(g[f"e{e}_scale_abs"][k])
                ref_mag = max(abs(want), sabs)
                es = abs(float(got) - want) / ref_mag
                self_rel = abs(float(got) - want) / abs(want)
                worst["scale"] = max(worst.get("scale", 0.0), es)
                worst["scale_rel_to_itself"] = max(worst.get("scale_rel_to_itself", 0.0), self_rel)
                assert es < tol, ("scale", e, k, got, want, ref_mag)
                # WELL-CONDITIONED scale edges (|s| >= 0.5 x the same ratio without cancellation: every edge of the full-architecture
                # sequences) meet the bar relative to the value ITSELF; only an edge whose numerator cancels by more than half - tiny
                # configuration, random two-signed depths - is exempt from that second assertion
                if abs(want) >= 0.5 * sabs:
                    worst["scale_well_conditioned"] = max(worst.get("scale_well_conditioned", 0.0), self_rel)
                    worst["n_well_conditioned"] = worst.get("n_well_conditioned", 0.0) + 1.0
                    assert self_rel < tol, ("scale relative to itself (well-conditioned edge)", e, k, got, want, sabs)
                else:
                    worst["n_ill_conditioned"] = worst.get("n_ill_conditioned", 0.0) + 1.0
                note("scale_conf", np.array([r["scale_confs"][k]]), np.array([float(g[f"e{e}_scale_conf"][k])]))
    assert worst.get("confs_norm", 0.0) < tol and worst.get("depths_norm", 0.0) < tol, worst
    return worst
