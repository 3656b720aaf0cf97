"""Pair-level data parallelism for the STA frontend: one process per GPU, `torch.distributed`
(backend "nccl" == RCCL over xGMI on the GPU box, "gloo" in the CPU tests).

Image pairs are independent units: weights are replicated, pairs are sharded
contiguously across ranks, there is NO collective on the compute path.  The only exchange is one
all-gather per step of the compact per-pair outputs a SLAM consumer reads - pose 4x4, pose
confidence, depth (= pts3d[...,2]) and the confidence map for both views (slam.py:165-185).
"""
from __future__ import annotations

from typing import Dict, List, Tuple

import torch
import torch.distributed as dist


def shard_range(num_pairs: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous, balanced [lo, hi) slice of the pair list owned by `rank` (ragged tail allowed)."""
    base, rem = divmod(num_pairs, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def compact_elems_per_pair(H: int, W: int) -> int:
    return 2 * (16 + 1 + 2 * H * W)


def pack_compact(main: Dict[str, torch.Tensor], supp: Dict[str, torch.Tensor], model=None,
                 out: torch.Tensor | None = None) -> torch.Tensor:
    """[B, compact_elems_per_pair] fp32: per pair (main then support) pose16, pose_conf, depth, conf.
    With `model` (an STAFrontend) and contiguous GPU outputs the record is written by ONE kernel (sta_pack_compact),
    straight into `out` when given (e.g. the send buffer of the step's all-gather); otherwise a torch.cat of the slices."""
    pts = main["pts3d_pred"]
    keys = ("pts3d_pred", "conf", "relative_pose", "relative_pose_conf")

    def fast_ok():
        """The one-kernel path reads raw fp32 device pointers: every input (and `out`) must be fp32, contiguous, on the same GPU
        and of the expected shape - anything else takes the torch path below, which converts or rejects it."""
        if model is None or not pts.is_cuda or pts.dim() != 4:
            return False
        B, H, W = pts.shape[0], pts.shape[1], pts.shape[2]
        want = {"pts3d_pred": (B, H, W, 3), "conf": (B, H, W), "relative_pose": (B, 4, 4), "relative_pose_conf": (B,)}
        for o in (main, supp):
            for k in keys:
                t = o[k]
                if t.dtype != torch.float32 or t.device != pts.device or not t.is_contiguous() or tuple(t.shape) != want[k]:
                    return False
        if out is not None and (out.dtype != torch.float32 or out.device != pts.device or out.dim() != 2):
            return False
        return True

    if fast_ok():
        import ctypes as C
        from . import _lib
        B, H, W = pts.shape[0], pts.shape[1], pts.shape[2]
        if out is None:
            out = torch.empty(B, compact_elems_per_pair(H, W), device=pts.device, dtype=torch.float32)
        assert out.shape[0] == B and out.stride(1) == 1 and out.shape[1] == compact_elems_per_pair(H, W)
        arr = [(C.c_void_p * 2)(main[k].data_ptr(), supp[k].data_ptr()) for k in keys]
        _lib.check(model.lib.sta_pack_compact(model._h, arr[0], arr[1], arr[2], arr[3], B, H, W, out.data_ptr(), out.stride(0),
                                              torch.cuda.current_stream(pts.device).cuda_stream))
        return out
    parts = []
    for o in (main, supp):
        B = o["relative_pose"].shape[0]
        parts += [o["relative_pose"].reshape(B, 16), o["relative_pose_conf"].reshape(B, 1),
                  o["pts3d_pred"][..., 2].reshape(B, -1), o["conf"].reshape(B, -1)]
    res = torch.cat([p.float() for p in parts], dim=1).contiguous()
    if out is not None:
        out.copy_(res)
        return out
    return res


def unpack_compact(buf: torch.Tensor, H: int, W: int) -> List[Dict[str, torch.Tensor]]:
    """Inverse of pack_compact -> [main, supp] dicts with pose [B,4,4], pose_conf [B], depth/conf [B,H,W]."""
    B = buf.shape[0]
    out, o = [], 0
    for _ in range(2):
        d = {"relative_pose": buf[:, o:o + 16].reshape(B, 4, 4)}; o += 16
        d["relative_pose_conf"] = buf[:, o]; o += 1
        d["depth"] = buf[:, o:o + H * W].reshape(B, H, W); o += H * W
        d["conf"] = buf[:, o:o + H * W].reshape(B, H, W); o += H * W
        out.append(d)
    return out


def gather_compact(local: torch.Tensor, num_pairs: int, out: torch.Tensor | None = None) -> torch.Tensor:
    """All-gather the per-rank compact rows into the global pair order on every rank.

    Equal shards use one all_gather_into_tensor (the steady-state bench path); ragged shards pad to
    the largest shard and strip the padding afterwards."""
    if not (dist.is_available() and dist.is_initialized()):
        return local
    world, rank = dist.get_world_size(), dist.get_rank()     # (a world of 1 still runs the collective: the GPU test uses it)
    sizes = [shard_range(num_pairs, world, r) for r in range(world)]
    counts = [hi - lo for lo, hi in sizes]
    mx = max(counts)
    width = local.shape[1]
    if min(counts) == mx:
        if out is None:
            out = torch.empty(world * mx, width, device=local.device, dtype=local.dtype)
        dist.all_gather_into_tensor(out, local)
        return out
    pad = torch.zeros(mx, width, device=local.device, dtype=local.dtype)
    pad[:counts[rank]] = local
    tmp = torch.empty(world * mx, width, device=local.device, dtype=local.dtype)
    dist.all_gather_into_tensor(tmp, pad)
    return torch.cat([tmp[r * mx: r * mx + counts[r]] for r in range(world)], dim=0)


# ---------------------------------------------------------------------------------------------------------
# SLAM-side partitioning: the <= neighbor_edge_num + loop_edge_num candidate edges of ONE keyframe are
# scattered over the ranks; every rank holds the replicated encoder features (<= 400 x 0.8 MB @224x224), runs the
# keyframe scheduler (`slam_scheduler.regress_views`) on its own edges, and one all-gather returns every edge's compact
# result (what `connect_view_i_j` consumes: pose, confidence, accepted flag, shared intrinsics, depths, conf maps).

def edge_shard(num_edges: int, world: int, rank: int) -> List[int]:
    """Edge indices owned by `rank`: round-robin, so the adjacent (always accepted, DPT-heavy) edge and the cheap
    rejected loop candidates spread evenly instead of piling on one rank."""
    return list(range(rank, num_edges, world))


def edge_elems(H: int, W: int) -> int:
    """accepted flag, pose_conf, pose 16, K 9, depths 2HW, confs 2HW"""
    return 2 + 16 + 9 + 4 * H * W


def pack_edges(results, H: int, W: int, device=None) -> torch.Tensor:
    """`results`: list of slam_scheduler.EdgeResult (or objects with the same fields) of THIS rank's edges.
    H, W = the dims of the maps AS RETURNED: for portrait frames regress_views hands out transposed views ([2, W_img, H_img],
    like the reference, utils/misc.py:60-61,81), so pass (W_img, H_img) here and to gather_edges - a mismatch would scramble the
    gathered maps silently, hence the check."""
    rows = []
    for r in results:
        if r.accepted and (tuple(r.depths.shape[-2:]) != (H, W) or tuple(r.confs.shape[-2:]) != (H, W)):
            raise ValueError(f"pack_edges(H={H}, W={W}) got maps of shape {tuple(r.depths.shape)} / {tuple(r.confs.shape)}: "
                             "pass the dims of the returned maps (transposed for portrait frames)")
        dev = r.pose.device if device is None else device
        head = torch.tensor([1.0 if r.accepted else 0.0, float(r.rel_pose_conf)], device=dev)
        if r.accepted:
            body = [r.pose.reshape(16), r.intri.reshape(9), r.depths.reshape(-1), r.confs.reshape(-1)]
        else:
            body = [r.pose.reshape(16), torch.zeros(9 + 4 * H * W, device=dev)]
        rows.append(torch.cat([head] + [b.to(dev, torch.float32) for b in body]))
    if not rows:
        return torch.empty(0, edge_elems(H, W), device=device)
    return torch.stack(rows).contiguous()


def gather_edges(local: torch.Tensor, num_edges: int, H: int, W: int) -> List[dict]:
    """All ranks -> the results of ALL edges in the original edge order (list of dicts; depths / confs / intri are None
    for rejected edges, exactly like regress_two_views' early return, slam.py:170)."""
    on = dist.is_available() and dist.is_initialized()
    world = dist.get_world_size() if on else 1
    width = edge_elems(H, W)
    per = (num_edges + world - 1) // world
    if on:
        pad = torch.zeros(per, width, device=local.device, dtype=torch.float32)
        pad[:local.shape[0]] = local
        allr = torch.empty(world * per, width, device=local.device, dtype=torch.float32)
        dist.all_gather_into_tensor(allr, pad)
    else:
        allr = local
    out: List[dict] = [None] * num_edges
    for r in range(world):
        for slot, e in enumerate(edge_shard(num_edges, world, r)):
            row = allr[r * per + slot]
            acc = bool(row[0] > 0.5)
            d = {"accepted": acc, "rel_pose_conf": float(row[1]), "pose": row[2:18].reshape(4, 4),
                 "intri": None, "depths": None, "confs": None}
            if acc:
                d["intri"] = row[18:27].reshape(3, 3)
                d["depths"] = row[27:27 + 2 * H * W].reshape(2, H, W)
                d["confs"] = row[27 + 2 * H * W:].reshape(2, H, W)
            out[e] = d
    return out
