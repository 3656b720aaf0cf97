"""f3 input step, CPU side: the oracle restatement of the reference crop / Pillow-LANCZOS / crop / ImgNorm / ImgGray
chain against goldens made by the real Pillow + the reference's own cropping module (oracle/gen_golden_pre.py)."""
import glob
import os

import numpy as np
import pytest

from oracle import preprocess_oracle as P
from vista_slam_amd import weights as W

GOLD = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "pre_*.npz")))


def test_goldens_present():
    assert len(GOLD) >= 6


@pytest.mark.parametrize("path", GOLD, ids=[os.path.basename(p)[4:-4] for p in GOLD])
def test_oracle_is_bit_exact_to_pillow_and_reference_cropping(path):
    g = np.load(path)
    Hs, Ws = (int(v) for v in g["src_hw"])
    ow, oh = (int(v) for v in g["target_wh"])
    src = W.synth_frames_u8(Hs, Ws, seed=43, tag=int(g["tag"]))
    assert int(src.astype(np.int64).sum()) == int(g["src_sum"])          # the procedural source frame is reproducible
    r = P.process_image(src, ow, oh)
    assert np.array_equal(r["u8"], g["u8"])                               # bit-exact resize + crops
    assert np.array_equal(r["rgb"][:, ::5, ::5], g["rgb_s"])
    assert np.array_equal(r["gray"][:, ::5, ::5], g["gray_s"])
    assert abs(float(r["rgb"].astype(np.float64).sum()) - float(g["rgb_sum"])) < 1e-6 * r["rgb"].size
    assert abs(float(r["gray"].astype(np.float64).sum()) - float(g["gray_sum"])) < 1e-6 * r["gray"].size


def test_geometry_tum_case():
    """640x480 with 10 px edges -> 620x460 crop -> 301x224 rescale -> columns 38..262 (np.round half-to-even)."""
    crop, (rw, rh), (l2, t2) = P.crop_resize_geometry(480, 640, 224, 224, 10, 10)
    assert crop == (10, 10, 630, 470) and (rw, rh) == (301, 224) and (l2, t2) == (38, 0)


def test_portrait_and_ambiguous_square_are_rejected():
    with pytest.raises(AssertionError):
        P.crop_resize_geometry(640, 480, 224, 224, 10, 10)
    with pytest.raises(AssertionError):
        P.crop_resize_geometry(500, 500, 512, 384, 10, 10)


def test_identity_size_is_identity():
    img = W.synth_frames_u8(64, 96, seed=43, tag=9)
    assert np.array_equal(P.lanczos_resize_u8(img, 96, 64), img)
