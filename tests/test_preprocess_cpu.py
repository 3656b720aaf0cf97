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
    assert len(GOLD) >= 9


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
    crop, (rw, rh), (l2, t2), out = P.crop_resize_geometry(480, 640, 224, 224, 10, 10)
    assert crop == (10, 10, 630, 470) and (rw, rh) == (301, 224) and (l2, t2) == (38, 0) and out == (224, 224)


def test_portrait_transposes_the_resolution_and_ambiguous_square_is_rejected():
    """base_view_graph_dataset.py:200-209: a portrait crop swaps (w, h); a square crop with a non-square resolution draws
    the orientation from an rng in the reference and is refused here; a portrait RESOLUTION is refused like the reference's assert."""
    assert P.crop_resize_geometry(640, 480, 512, 384, 10, 10)[3] == (384, 512)
    assert P.crop_resize_geometry(640, 480, 224, 224, 10, 10)[3] == (224, 224)
    with pytest.raises(AssertionError):
        P.crop_resize_geometry(500, 500, 512, 384, 10, 10)
    with pytest.raises(AssertionError):
        P.crop_resize_geometry(480, 640, 384, 512, 10, 10)


def test_identity_size_is_identity():
    img = W.synth_frames_u8(64, 96, seed=43, tag=9)
    assert np.array_equal(P.lanczos_resize_u8(img, 96, 64), img)


def test_oracle_lanczos_equals_live_pillow_on_random_geometries():
    """Beyond the committed goldens: the restated resampler against the Pillow installed in this image
    (a pip dependency of the reference, not reference code) on random up / down-scaling geometries."""
    PIL = pytest.importorskip("PIL.Image")
    rng = np.random.default_rng(11)
    for it in range(10):
        Hs, Ws = int(rng.integers(20, 300)), int(rng.integers(20, 400))
        oh, ow = int(rng.integers(8, 260)), int(rng.integers(8, 330))
        img = W.synth_frames_u8(Hs, Ws, seed=43, tag=100 + it)
        want = np.asarray(PIL.fromarray(img).resize((ow, oh), resample=PIL.Resampling.LANCZOS))
        got = P.lanczos_resize_u8(img, ow, oh)
        assert np.array_equal(got, want), (Hs, Ws, oh, ow)


def test_ply_writer_round_trip(tmp_path):
    """f4 host formatting (no GPU): header + 27-byte records survive a write / read cycle."""
    from vista_slam_amd import formats as F
    rec = np.zeros(5, dtype=F.PLY_RECORD)
    rec["x"] = np.arange(5) * 0.5; rec["y"] = -np.arange(5); rec["z"] = 1e-3
    rec["red"] = [0, 1, 127, 254, 255]; rec["green"] = 7; rec["blue"] = 9
    path = str(tmp_path / "c.ply")
    F.write_ply(path, rec)
    head = open(path, "rb").read(200).decode("ascii", "ignore")
    assert head.startswith("ply\nformat binary_little_endian 1.0\n") and "element vertex 5\n" in head and "property double x" in head
    assert np.array_equal(F.read_ply(path), rec)
