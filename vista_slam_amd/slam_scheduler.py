"""Keyframe scheduler behind the reference's `regress_two_views` contract.

`OnlineSLAM.step` (vista_slam/slam.py:263-277) connects a new keyframe i to its <= neighbor_edge_num previous views
and <= loop_edge_num loop candidates by calling `regress_two_views(i, j)` (slam.py:153-189) once per edge: a B=1
decode, the pose head, an early return for low-confidence non-adjacent edges, then two DPT heads, the shared
intrinsics and the depths.  `regress_views` does the same for all candidate edges in one native call
(`sta_regress_views`, include/sta_mi355.h): one batched decode, pose heads first, one D2H read of the k
confidences, DPT + reductions only for the accepted edges.  All arithmetic runs in libsta_mi355.so.
"""
from __future__ import annotations

import ctypes as C
from typing import List, Optional, Sequence

import torch

from . import _lib
from .sta_frontend import STAFrontend, _stream_ptr


class EdgeResult:
    """What `regress_two_views` returns for one edge (slam.py:170,189): the relative pose (4x4; the reference converts
    it with pp.mat2SE3, see `formats.mat_to_se3`), its confidence, and - for accepted edges - confs [2,H,W],
    intri [3,3], depths [2,H,W] (+ the full point maps, which the reference discards after the reductions)."""
    __slots__ = ("pose", "rel_pose_conf", "accepted", "confs", "intri", "depths", "pts3d")

    def __init__(self, pose, rel_pose_conf, accepted, confs=None, intri=None, depths=None, pts3d=None):
        self.pose, self.rel_pose_conf, self.accepted = pose, rel_pose_conf, accepted
        self.confs, self.intri, self.depths, self.pts3d = confs, intri, depths, pts3d

    def as_reference_tuple(self):
        """(pose_ij 4x4, rel_pose_conf_ij, confs, intri, depths) with None for rejected edges (slam.py:170)."""
        return self.pose, self.rel_pose_conf, self.confs, self.intri, self.depths


def regress_views(frontend: STAFrontend, enc_feat_i: torch.Tensor, enc_feats_j: Sequence[torch.Tensor],
                  adjacent: Sequence[bool], rel_pose_thres: float, H: int, W: int) -> List[EdgeResult]:
    """Edges (i, j_e), e < k, of one keyframe.  enc_feat_i / enc_feats_j[e]: [1,N,1024] encoder features as cached
    by `add_view` (slam.py:142-151).  adjacent[e] = (i - j_e == 1).  Synchronises the current stream once."""
    k = len(enc_feats_j)
    assert k == len(adjacent) and 1 <= k <= 16
    frontend._check_hw(H, W)
    dev = frontend.device
    N, E = (H // 16) * (W // 16), frontend.cfg.enc_embed_dim
    fi = enc_feat_i.to(dev, torch.float32).contiguous()
    fj = [f.to(dev, torch.float32).contiguous() for f in enc_feats_j]
    for f in [fi] + fj:
        assert f.numel() == N * E, f"encoder feature has {f.numel()} elements, expected {N}x{E}"
    ptrs = (C.c_void_p * k)(*[f.data_ptr() for f in fj])
    adj = bytes(bytearray(1 if a else 0 for a in adjacent))
    pose = torch.empty(k, 4, 4, device=dev, dtype=torch.float32)
    pts = torch.empty(k, 2, H, W, 3, device=dev, dtype=torch.float32)
    conf = torch.empty(k, 2, H, W, device=dev, dtype=torch.float32)
    Kt = torch.empty(k, 3, 3, device=dev, dtype=torch.float32)
    depth = torch.empty(k, 2, H, W, device=dev, dtype=torch.float32)
    pconf = (C.c_float * k)()
    slot = (C.c_int * k)()
    nacc = C.c_int(0)
    _lib.check(frontend.lib.sta_regress_views(frontend._h, fi.data_ptr(), ptrs, k, adj, float(rel_pose_thres), H, W,
                                              pose.data_ptr(), pconf, slot, C.byref(nacc), pts.data_ptr(), conf.data_ptr(),
                                              Kt.data_ptr(), depth.data_ptr(), frontend._stream()))
    if H > W:     # portrait: the reference sees transposed views of the same memory (utils/misc.py:60-61,81)
        pts, conf, depth = pts.swapaxes(2, 3), conf.swapaxes(2, 3), depth.swapaxes(2, 3)
    out = []
    for e in range(k):
        s = slot[e]
        if s < 0:
            out.append(EdgeResult(pose[e], float(pconf[e]), False))
        else:
            out.append(EdgeResult(pose[e], float(pconf[e]), True, conf[s], Kt[s], depth[s], pts[s]))
    return out
