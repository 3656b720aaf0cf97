"""Input step behind the reference's `process_image` contract.

`SLAM_image_only.process_image(rgb_image, img_name)` (vista_slam/datasets/slam_images_only.py:19-33) crops / LANCZOS-
rescales / centre-crops a uint8 RGB frame on the CPU with Pillow and applies the torchvision transforms ImgNorm and
ImgGray.  `process_image` here does the same on the GPU in two fused kernels of libsta_mi355.so
(`sta_preprocess_frame`, include/sta_mi355.h) and additionally returns the uint8 HWC frame, which
`STAFrontend.encode_u8hwc` consumes directly (coalesced HWC load, normalisation fused into the patch gather).
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _lib
from .sta_frontend import STAFrontend, _stream_ptr


def process_image(frontend: STAFrontend, rgb_image, resolution=(224, 224), w_edge: int = 10, h_edge: int = 10,
                  img_name: str | None = None) -> dict:
    """rgb_image: [Hs,Ws,3] uint8 (numpy or torch, host or device).  resolution = (width, height) like the reference.
    -> {'rgb': [3,H,W] fp32 in [-1,1], 'gray': [1,H,W] fp32, 'u8': [H,W,3] uint8, 'img_name': ...} on the device."""
    if isinstance(rgb_image, np.ndarray):
        rgb_image = torch.from_numpy(np.ascontiguousarray(rgb_image))
    assert rgb_image.dtype == torch.uint8 and rgb_image.dim() == 3 and rgb_image.shape[2] == 3, "expected [H,W,3] uint8"
    src = rgb_image.to(frontend.device).contiguous()
    Hs, Ws, _ = src.shape
    rw, rh = int(resolution[0]), int(resolution[1])
    oh_, ow_ = C.c_int(0), C.c_int(0)     # the resolution, transposed for a portrait frame (base_view_graph_dataset.py:200-205)
    _lib.check(frontend.lib.sta_preprocess_geometry(Hs, Ws, rh, rw, int(w_edge), int(h_edge), C.byref(oh_), C.byref(ow_)))
    oh, ow = oh_.value, ow_.value
    u8 = torch.empty(oh, ow, 3, device=frontend.device, dtype=torch.uint8)
    rgb = torch.empty(3, oh, ow, device=frontend.device, dtype=torch.float32)
    gray = torch.empty(1, oh, ow, device=frontend.device, dtype=torch.float32)
    _lib.check(frontend.lib.sta_preprocess_frame(frontend._h, src.data_ptr(), Hs, Ws, rh, rw, int(w_edge), int(h_edge),
                                                 u8.data_ptr(), rgb.data_ptr(), gray.data_ptr(), frontend._stream()))
    out = {"rgb": rgb, "gray": gray, "u8": u8}
    if img_name is not None:
        import os.path as osp
        out["img_name"] = osp.basename(img_name)
    return out
