"""Golden vectors for the input step, generated HERE from the real Pillow and the real
reference `vista_slam.utils.cropping` module (imported from /root/reference with a cv2 stub - cv2 is only used for
depth maps, which this path does not have).  The thin glue of `_crop_resize_if_necessary_image_only`
(base_view_graph_dataset.py:171-225) cannot be imported (its module needs torchvision / networkx), so its bbox
arithmetic is re-expressed below around the reference's own crop / rescale functions.

    python oracle/gen_golden_pre.py         # -> tests/golden/pre_*.npz
"""
import os
import sys
import types

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference")
sys.modules.setdefault("cv2", types.ModuleType("cv2"))

import PIL.Image  # noqa: E402
from vista_slam.utils import cropping  # noqa: E402  (reference code, imported, never copied)

from vista_slam_amd import weights as W  # noqa: E402

CASES = [  # (name, source H, W, target (w, h), tag)
    ("tum_480x640_to_224", 480, 640, (224, 224), 0),
    ("tum_480x640_to_512x384", 480, 640, (512, 384), 1),
    ("scenes_968x1296_to_224", 968, 1296, (224, 224), 2),
    ("kitti_376x1241_to_512x384", 376, 1241, (512, 384), 3),
    ("upscale_250x330_to_224", 250, 330, (224, 224), 4),
    ("odd_487x651_to_224", 487, 651, (224, 224), 5),
    # portrait frames: the resolution is transposed (base_view_graph_dataset.py:200-205)
    ("portrait_640x480_to_512x384", 640, 480, (512, 384), 6),
    ("portrait_1296x968_to_224", 1296, 968, (224, 224), 7),
    ("portrait_700x300_to_512x384", 700, 300, (512, 384), 8),
]


def reference_process(image_np, resolution, w_edge=10, h_edge=10):
    image = PIL.Image.fromarray(image_np)
    Wd, Hd = image.size
    cx, cy = int(Wd / 2), int(Hd / 2)
    mx, my = min(cx, Wd - cx), min(cy, Hd - cy)
    l, t, r, b = cx - mx, cy - my, cx + mx, cy + my
    l, t = max(l, w_edge), max(t, h_edge)
    r, b = min(r, Wd - w_edge), min(b, Hd - h_edge)
    image, _, _ = cropping.crop_image_depthmap(image, None, None, (l, t, r, b))
    W1, H1 = image.size
    assert resolution[0] >= resolution[1]
    if H1 > 1.1 * W1:                       # portrait: transposed resolution (:203-205)
        resolution = resolution[::-1]
    assert not (0.9 < H1 / W1 < 1.1 and resolution[0] != resolution[1]), "rng branch (:206-209) not covered"
    target = np.array(resolution)
    image, _, _ = cropping.rescale_image_depthmap(image, None, None, target)
    cw, ch = image.size
    ow, oh = target
    l2, t2 = np.int32(np.round(cw / 2 - ow / 2)), np.int32(np.round(ch / 2 - oh / 2))
    image, _, _ = cropping.crop_image_depthmap(image, None, None, (l2, t2, l2 + ow, t2 + oh))
    return np.asarray(image)


def main():
    import torch
    out_dir = os.path.join(ROOT, "tests", "golden")
    for name, Hs, Ws, res, tag in CASES:
        src = W.synth_frames_u8(Hs, Ws, seed=43, tag=tag)
        u8 = reference_process(src, res)
        assert u8.shape in ((res[1], res[0], 3), (res[0], res[1], 3)) and u8.dtype == np.uint8
        # torchvision is absent: ToTensor / Normalize(0.5,0.5) / Grayscale(1) from their definitions, in torch fp32
        t = torch.from_numpy(u8).permute(2, 0, 1).to(torch.float32).div(255)
        rgb = t.clone().sub_(0.5).div_(0.5)
        gray = (0.2989 * t[0] + 0.587 * t[1] + 0.114 * t[2]).unsqueeze(0)
        np.savez_compressed(os.path.join(out_dir, f"pre_{name}.npz"),
                            src_hw=np.array([Hs, Ws]), target_wh=np.array(res), tag=np.array(tag),
                            src_sum=np.array(int(src.astype(np.int64).sum())),
                            u8=u8, rgb_s=rgb.numpy()[:, ::5, ::5], gray_s=gray.numpy()[:, ::5, ::5],
                            rgb_sum=np.array(float(rgb.double().sum())), gray_sum=np.array(float(gray.double().sum())))
        print(name, u8.shape, int(u8.astype(np.int64).sum()))


if __name__ == "__main__":
    main()
