"""Post-STA reductions behind the reference's own function signatures.

`estimate_intrinsic_from_pts3d` and `estimate_scale_with_depth_and_confidence` mirror
vista_slam/utils/slam_utils.py:8-79 and :168-190 (called from slam.py:184,224-227); the fused variant
`pair_reductions` also returns depths = pts[...,2] (slam.py:185) and the per-image mean confidence
(pose_graph.py:37) from the same pass over the point maps.  All arithmetic runs in libsta_mi355.so.
"""
from __future__ import annotations

import torch

from . import _lib
from .sta_frontend import STAFrontend, _stream_ptr


def _prep(t: torch.Tensor) -> torch.Tensor:
    assert t.is_cuda, "post-STA reductions run on the GPU tensors the frontend returned"
    return t.float().contiguous()


def pair_reductions(frontend: STAFrontend, pts3d: torch.Tensor, confidence: torch.Tensor, shared_intrinsic: bool = False):
    """-> (K, depths [B,H,W], conf_mean [B]) in one HBM pass."""
    pts3d, confidence = _prep(pts3d), _prep(confidence)
    B, H, W, _ = pts3d.shape
    K = torch.empty((3, 3) if shared_intrinsic else (B, 3, 3), device=pts3d.device, dtype=torch.float32)
    depth = torch.empty(B, H, W, device=pts3d.device, dtype=torch.float32)
    cmean = torch.empty(B, device=pts3d.device, dtype=torch.float32)
    _lib.check(frontend.lib.sta_estimate_intrinsics(frontend._h, pts3d.data_ptr(), confidence.data_ptr(), B, H, W,
                                                    int(shared_intrinsic), K.data_ptr(), depth.data_ptr(), cmean.data_ptr(),
                                                    frontend._stream()))
    return K, depth, cmean


def estimate_intrinsic_from_pts3d(frontend: STAFrontend, pts3d: torch.Tensor, confidence: torch.Tensor,
                                  shared_intrinsic: bool = False) -> torch.Tensor:
    pts3d, confidence = _prep(pts3d), _prep(confidence)
    B, H, W, _ = pts3d.shape
    K = torch.empty((3, 3) if shared_intrinsic else (B, 3, 3), device=pts3d.device, dtype=torch.float32)
    _lib.check(frontend.lib.sta_estimate_intrinsics(frontend._h, pts3d.data_ptr(), confidence.data_ptr(), B, H, W,
                                                    int(shared_intrinsic), K.data_ptr(), None, None, frontend._stream()))
    return K


def estimate_scale_with_depth_and_confidence(frontend: STAFrontend, Di, Dj, ci, cj) -> torch.Tensor:
    Di, Dj, ci, cj = (_prep(t).reshape(-1) for t in (Di, Dj, ci, cj))
    s = torch.empty((), device=Di.device, dtype=torch.float32)
    _lib.check(frontend.lib.sta_estimate_scale(frontend._h, Di.data_ptr(), Dj.data_ptr(), ci.data_ptr(), cj.data_ptr(),
                                               Di.numel(), s.data_ptr(), frontend._stream()))
    return s
