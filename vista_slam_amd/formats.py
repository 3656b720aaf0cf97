"""Output step: the files `OnlineSLAM.save_data_all` writes (vista_slam/slam.py:338-421)
and the pose conversion of `regress_two_views` (slam.py:166), behind the same names, keys, dtypes and shapes, so
`eval/eval_recon.load_data` (vista_slam/eval/eval_recon.py:7-35) and `eval/eval_traj` read them unchanged.

Files: `trajectory[_postfix].npy [N,4,4]`, `scales[_postfix].npy [N,1]`, `images.npy [N,H,W,3]` in [0,1],
`depths.npy [N,H,W]` (unscaled), `confs.npz {confs [N,H,W], thres}`, `intrinsics.npy [N,3,3]`,
`view_graph.npz {view_graph (pickled dict), loop_min_dist, view_names}`, `pointcloud.ply`, `gt_*.npy`.
The only arithmetic (world point cloud, mat -> SE3) runs in libsta_mi355.so; the rest is formatting.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch

from . import _lib
from .sta_frontend import STAFrontend, _stream_ptr

PLY_RECORD = np.dtype([("x", "<f8"), ("y", "<f8"), ("z", "<f8"), ("red", "u1"), ("green", "u1"), ("blue", "u1")])
assert PLY_RECORD.itemsize == 27


def _dev(frontend, t, shape=None):
    t = torch.as_tensor(t).to(frontend.device, torch.float32).contiguous()
    if shape is not None:
        assert tuple(t.shape) == tuple(shape), f"expected shape {tuple(shape)}, got {tuple(t.shape)}"
    return t


def mat_to_se3(frontend: STAFrontend, pose: torch.Tensor) -> torch.Tensor:
    """pp.mat2SE3(pose).data (slam.py:166): [B,4,4] -> [B,7] = (tx,ty,tz,qx,qy,qz,qw)."""
    p = _dev(frontend, pose).reshape(-1, 4, 4)
    out = torch.empty(p.shape[0], 7, device=frontend.device, dtype=torch.float32)
    _lib.check(frontend.lib.sta_mat_to_se3(frontend._h, p.data_ptr(), p.shape[0], out.data_ptr(), frontend._stream()))
    return out


def world_pointcloud(frontend: STAFrontend, depths, scales, intrinsics, poses, confs, imgs, conf_thres: float,
                     want_records: bool = False):
    """slam.py:396-408 -> (points [M,3] fp32, colors [M,3] fp32 [, records [M] PLY_RECORD numpy array])."""
    depths = _dev(frontend, depths)
    N, H, W = depths.shape
    scales = _dev(frontend, scales).reshape(N)
    K = _dev(frontend, intrinsics, (N, 3, 3))
    poses = _dev(frontend, poses, (N, 4, 4))
    confs = _dev(frontend, confs, (N, H, W))
    imgs = _dev(frontend, imgs, (N, 3, H, W)) if imgs is not None else None
    cap = N * H * W
    pts = torch.empty(cap, 3, device=frontend.device, dtype=torch.float32)
    col = torch.empty(cap, 3, device=frontend.device, dtype=torch.float32)
    rec = torch.empty(cap * 27, device=frontend.device, dtype=torch.uint8) if want_records else None
    cnt = C.c_int64(0)
    _lib.check(frontend.lib.sta_world_pointcloud(frontend._h, depths.data_ptr(), scales.data_ptr(), K.data_ptr(),
                                                 poses.data_ptr(), confs.data_ptr(),
                                                 imgs.data_ptr() if imgs is not None else None, N, H, W, float(conf_thres),
                                                 pts.data_ptr(), col.data_ptr(), rec.data_ptr() if rec is not None else None,
                                                 C.byref(cnt), frontend._stream()))
    M = cnt.value
    if want_records:
        records = np.frombuffer(rec[:M * 27].cpu().numpy().tobytes(), dtype=PLY_RECORD)
        return pts[:M], col[:M], records
    return pts[:M], col[:M]


def write_ply(path: str, records: np.ndarray):
    """Binary little-endian PLY of a coloured cloud with double coordinates - the layout Open3D's
    `write_point_cloud` produces for slam.py:405-408 (readable by `o3d.io.read_point_cloud`)."""
    assert records.dtype == PLY_RECORD
    header = ("ply\nformat binary_little_endian 1.0\ncomment Created by vista_slam_amd (Open3D layout)\n"
              f"element vertex {len(records)}\nproperty double x\nproperty double y\nproperty double z\n"
              "property uchar red\nproperty uchar green\nproperty uchar blue\nend_header\n")
    with open(path, "wb") as f:
        f.write(header.encode("ascii"))
        f.write(records.tobytes())


def read_ply(path: str) -> np.ndarray:
    with open(path, "rb") as f:
        n = None
        while True:
            line = f.readline().decode("ascii").strip()
            if line.startswith("element vertex"):
                n = int(line.split()[-1])
            if line == "end_header":
                break
        return np.frombuffer(f.read(n * 27), dtype=PLY_RECORD)


def save_data_all(frontend: STAFrontend, output_folder: str, *, poses, scales, depths, confs, intrinsics, imgs,
                  conf_thres: float, view_graph: Optional[Dict[int, List[int]]] = None, loop_min_dist=None,
                  view_names: Optional[Sequence[str]] = None, save_view_graph=True, traj_name_postfix=None,
                  save_poses=True, save_images=True, save_scales=True, save_depths=True, save_intrinsics=True,
                  save_confs=True, save_ply=True, gt_poses=None, gt_depths=None, gt_intrinsics=None):
    """Same switches and files as OnlineSLAM.save_data_all (slam.py:338-421).  poses [N,4,4] (rotation + translation
    of the best node's Sim3), scales [N,1], depths / confs [N,H,W], intrinsics [N,3,3], imgs [N,3,H,W] in [-1,1]."""
    os.makedirs(output_folder, exist_ok=True)

    def host(t):
        return torch.as_tensor(t).detach().cpu().numpy()
    if save_view_graph:
        np.savez(f"{output_folder}/view_graph.npz", view_graph=view_graph, loop_min_dist=loop_min_dist,
                 view_names=list(view_names) if view_names is not None else None)
    post = f"_{traj_name_postfix}" if traj_name_postfix is not None else ""
    if save_poses:
        np.save(f"{output_folder}/trajectory{post}.npy", host(poses))
    if save_scales:
        np.save(f"{output_folder}/scales{post}.npy", host(scales))
    if save_images:
        images = (torch.as_tensor(imgs).detach().cpu().float().permute(0, 2, 3, 1) + 1.0) / 2.0
        np.save(f"{output_folder}/images.npy", images.numpy())
    if save_depths:
        np.save(f"{output_folder}/depths.npy", host(depths))
    if save_confs:
        np.savez(f"{output_folder}/confs.npz", confs=host(confs), thres=conf_thres)
    if save_intrinsics:
        np.save(f"{output_folder}/intrinsics.npy", host(intrinsics))
    if save_ply:
        _, _, records = world_pointcloud(frontend, depths, scales, intrinsics, poses, confs, imgs, conf_thres, want_records=True)
        write_ply(f"{output_folder}/pointcloud.ply", records)
    if gt_poses is not None:
        np.save(f"{output_folder}/gt_poses.npy", np.array(gt_poses).astype(np.float32))
    if gt_depths is not None:
        np.save(f"{output_folder}/gt_depths.npy", np.array(gt_depths).astype(np.float32))
    if gt_intrinsics is not None:
        np.save(f"{output_folder}/gt_intrinsics.npy", gt_intrinsics)
