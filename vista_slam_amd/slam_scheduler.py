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
import threading
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


class PendingEdges:
    """A scheduler call between its two phases (sta_regress_views_begin / _finish): the output tensors, the inputs kept
    alive, and the stream the call lives on.  While it is open its stream's scratch context inside the library is reserved;
    `regress_views_finish` closes it, and so does `close()` / leaving a `with` block (an exception between the phases must not
    leave the stream unusable: sta_regress_views_abort).  Close it EXPLICITLY: `__del__` is only a best-effort fallback and acts
    only on the thread that created the object - the abort may block in hipEventSynchronize and the handle is not thread-safe,
    so a garbage collection that happens to run on another thread must not enter the library."""
    __slots__ = ("k", "H", "W", "stream", "pose", "pts", "conf", "K", "depth", "_keep", "_frontend", "_open", "_tid")

    def close(self):
        """Abort the call if it is still pending (idempotent)."""
        if getattr(self, "_open", False):
            self._open = False
            fe = self._frontend
            if getattr(fe, "_h", None):
                fe.lib.sta_regress_views_abort(fe._h, self.stream)

    def __enter__(self):
        return self

    def __exit__(self, *_exc):
        self.close()
        return False

    def __del__(self):
        try:
            if getattr(self, "_tid", None) == threading.get_ident():
                self.close()
        except Exception:
            pass


def regress_views_begin(frontend: STAFrontend, enc_feat_i: torch.Tensor, enc_feats_j: Sequence[torch.Tensor], H: int, W: int) -> PendingEdges:
    """Phase 1 of `regress_views` on the CURRENT stream: gather + batched decode + pose heads, no host synchronisation.
    Until `regress_views_finish` no other frontend call may run on this stream (calls on other streams are fine)."""
    k = len(enc_feats_j)
    assert 1 <= k <= 16
    frontend._check_hw(H, W)
    dev = frontend.device
    N, E = (H // 16) * (W // 16), frontend.cfg.enc_embed_dim
    fi = enc_feat_i.to(dev, torch.float32).contiguous()
    fj = [f.to(dev, torch.float32).contiguous() for f in enc_feats_j]
    for f in [fi] + fj:
        assert f.numel() == N * E, f"encoder feature has {f.numel()} elements, expected {N}x{E}"
    ptrs = (C.c_void_p * k)(*[f.data_ptr() for f in fj])
    p = PendingEdges()
    p.k, p.H, p.W, p.stream = k, H, W, frontend._stream()
    p.pose = torch.empty(k, 4, 4, device=dev, dtype=torch.float32)
    p.pts = torch.empty(k, 2, H, W, 3, device=dev, dtype=torch.float32)
    p.conf = torch.empty(k, 2, H, W, device=dev, dtype=torch.float32)
    p.K = torch.empty(k, 3, 3, device=dev, dtype=torch.float32)
    p.depth = torch.empty(k, 2, H, W, device=dev, dtype=torch.float32)
    p._keep = (fi, fj)
    p._frontend, p._open, p._tid = frontend, False, threading.get_ident()
    _lib.check(frontend.lib.sta_regress_views_begin(frontend._h, fi.data_ptr(), ptrs, k, H, W, p.pose.data_ptr(), p.stream))
    p._open = True
    return p


def regress_views_finish(frontend: STAFrontend, p: PendingEdges, adjacent: Sequence[bool], rel_pose_thres: float) -> List[EdgeResult]:
    """Phase 2: waits for the k pose confidences, accepts / rejects (slam.py:169), enqueues the DPT heads + reductions of the
    accepted edges on the stream `regress_views_begin` ran on."""
    k, H, W = p.k, p.H, p.W
    assert k == len(adjacent)
    adj = bytes(bytearray(1 if a else 0 for a in adjacent))
    pconf = (C.c_float * k)()
    slot = (C.c_int * k)()
    nacc = C.c_int(0)
    assert p._open, "this scheduler call was already finished or aborted"
    p._open = False
    rc = frontend.lib.sta_regress_views_finish(frontend._h, adj, float(rel_pose_thres), pconf, slot, C.byref(nacc),
                                               p.pts.data_ptr(), p.conf.data_ptr(), p.K.data_ptr(), p.depth.data_ptr(), p.stream)
    if rc != 0:
        # a failure BEFORE the C side cleared its pending flag (an argument check, a context lookup) would leave the stream
        # reserved until sta_destroy: abort unconditionally - idempotent, 0 when nothing is pending - and keep the first error
        msg = frontend.lib.sta_last_error()
        frontend.lib.sta_regress_views_abort(frontend._h, p.stream)
        raise _lib.StaError(msg.decode() if isinstance(msg, bytes) else str(msg))
    pts, conf, depth = p.pts, p.conf, p.depth
    if H > W:     # portrait: the reference sees transposed views of the same memory (utils/misc.py:60-61,81)
        pts, conf, depth = pts.swapaxes(2, 3), conf.swapaxes(2, 3), depth.swapaxes(2, 3)
    out = []
    for e in range(k):
        s = slot[e]
        if s < 0:
            out.append(EdgeResult(p.pose[e], float(pconf[e]), False))
        else:
            out.append(EdgeResult(p.pose[e], float(pconf[e]), True, conf[s], p.K[s], depth[s], pts[s]))
    return out


def regress_views(frontend: STAFrontend, enc_feat_i: torch.Tensor, enc_feats_j: Sequence[torch.Tensor],
                  adjacent: Sequence[bool], rel_pose_thres: float, H: int, W: int) -> List[EdgeResult]:
    """Edges (i, j_e), e < k, of one keyframe.  enc_feat_i / enc_feats_j[e]: [1,N,1024] encoder features as cached
    by `add_view` (slam.py:142-151).  adjacent[e] = (i - j_e == 1).  Synchronises once, on the k pose confidences."""
    assert len(enc_feats_j) == len(adjacent)
    return regress_views_finish(frontend, regress_views_begin(frontend, enc_feat_i, enc_feats_j, H, W), adjacent, rel_pose_thres)
