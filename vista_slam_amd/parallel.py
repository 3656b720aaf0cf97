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


def pack_compact(main: Dict[str, torch.Tensor], supp: Dict[str, torch.Tensor]) -> torch.Tensor:
    """[B, compact_elems_per_pair] fp32: per pair (main then support) pose16, pose_conf, depth, conf."""
    parts = []
    for o in (main, supp):
        B = o["relative_pose"].shape[0]
        parts += [o["relative_pose"].reshape(B, 16), o["relative_pose_conf"].reshape(B, 1),
                  o["pts3d_pred"][..., 2].reshape(B, -1), o["conf"].reshape(B, -1)]
    return torch.cat(parts, dim=1).contiguous()


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
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return local
    world, rank = dist.get_world_size(), dist.get_rank()
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
