"""CPU: pin the oracle (oracle/sta_oracle.py + sta_oracle_ops.c) against golden vectors generated
from the REFERENCE PyTorch model (oracle/gen_golden.py imports /root/reference in the build container).
The reference ships no tests of its own, so these fixtures are the pin."""
import numpy as np
import pytest

from helpers import load_golden, rel_l2, max_rel
from oracle import sta_oracle as O
from vista_slam_amd import weights as W

TOL = 2e-5          # fp32 reduction-order noise floor of the full forward (default weights)
TOL_SHARP = 2e-4    # peaky-attention sets amplify fp32 noise (oracle vs reference, both fp32)


@pytest.fixture(scope="module")
def ops():
    return load_golden("ops")[0]


def test_rope2d(ops):
    assert max_rel(O.rope2d(ops["rope_tok"], ops["rope_pos"], 100.0), ops["rope_out"]) < 2e-6


def test_layernorm_gelu(ops):
    assert max_rel(O.layernorm(ops["ln_x"], ops["ln_w"], ops["ln_b"], 1e-6), ops["ln_out"]) < 2e-6
    assert max_rel(O.gelu(ops["gelu_x"]), ops["gelu_out"]) < 2e-6


@pytest.mark.parametrize("k", [4, 2])
def test_conv_transpose(ops, k):
    y = O.conv_transpose2d(ops[f"convt{k}_x"], ops[f"convt{k}_w"], ops[f"convt{k}_b"], k)
    assert max_rel(y, ops[f"convt{k}_out"]) < 2e-6


@pytest.mark.parametrize("tag", ["odd", "even"])
def test_conv3x3_stride2(ops, tag):
    y = O.conv2d(ops[f"conv3s2_{tag}_x"], ops[f"conv3s2_{tag}_w"], ops[f"conv3s2_{tag}_b"], 2, 1)
    assert y.shape == ops[f"conv3s2_{tag}_out"].shape
    assert max_rel(y, ops[f"conv3s2_{tag}_out"]) < 2e-6


def test_bilinear_align_corners(ops):
    assert max_rel(O.bilinear_up2(ops["bilin_x"]), ops["bilin_out"]) < 2e-6


def test_svd_orthogonalize_incl_reflection(ops):
    r = O.svd_orthogonalize(ops["svd_in"])
    assert np.abs(r - ops["svd_out"]).max() < 5e-6
    assert np.allclose(np.linalg.det(r.astype(np.float64)), 1.0, atol=1e-5)


def test_postprocess(ops):
    pts, conf = O.postprocess(ops["post_in"])
    assert max_rel(pts, ops["post_pts"]) < 2e-6 and max_rel(conf, ops["post_conf"]) < 2e-6
    assert np.all(pts[0, 0, 0] == 0)            # zero-norm pixel: clip(1e-8) path


@pytest.mark.parametrize("case", ["tiny_32x32_b1", "tiny_48x64_b2", "tiny_48x64_b2_sharp", "tiny_48x80_smooth_sharp",
                                  "tiny_48x80_sharp_s44_smooth", "tiny_48x80_sharp_s45", "tiny_48x80_sharp_s46_smooth", "tiny_48x80_sharp_s47",
                                  "tiny_80x48_b2_portrait",
                                  # trained-checkpoint-like range statistics (weights.py _outlier): heavy-tailed LayerNorm gains,
                                  # massive activation channels, DPT feature maps 300x / 3e4x larger
                                  "tiny_48x64_b2_outlier", "tiny_48x80_outlier_sharp", "tiny_48x64_b1_overflow"])
def test_forward_vs_reference_golden(case):
    g, meta = load_golden(case)
    H, W_, B = int(meta["H"]), int(meta["W"]), int(meta["B"])
    sd = W.state_dict(W.TINY, seed=int(meta["seed"]), qk_gain=float(meta["qk_gain"]), outlier=int(meta.get("outlier", 0)))
    gen = W.smooth_images if int(meta["smooth"]) else W.synth_images
    imgs = gen(2 * B, H, W_, seed=int(meta["seed"]), tag=0)
    r = O.forward_pair(W.TINY, sd, imgs[:B], imgs[B:])
    tol = TOL_SHARP if float(meta["qk_gain"]) != 1.0 else TOL
    errs = {}
    for side in ("main", "supp"):
        errs[f"{side}_pts3d"] = rel_l2(r[side]["pts3d"], g[f"{side}_pts3d"])
        errs[f"{side}_conf"] = rel_l2(r[side]["conf"], g[f"{side}_conf"])
        errs[f"{side}_pose"] = rel_l2(r[side]["pose"], g[f"{side}_pose"])
        errs[f"{side}_pose_conf"] = rel_l2(r[side]["pose_conf"], g[f"{side}_pose_conf"])
    errs["enc_feat_a"] = rel_l2(r["enc_feat_a"], g["enc_feat_a"])
    # integer positions of _encode_image (PositionGetter, sta_blocks.py:241-247): bit-exact
    hp, wp = H // 16, W_ // 16
    pos = O.positions(B, hp, wp)
    assert pos.dtype == np.int64 and np.array_equal(pos, g["pos_a"]) and np.array_equal(pos, g["pos_b"])
    for hk in W.TINY.hooks[1:]:
        errs[f"dec1_hook{hk - 1}"] = rel_l2(r["dec1"][hk - 1], g[f"dec1_hook{hk - 1}"])
        errs[f"dec2_hook{hk - 1}"] = rel_l2(r["dec2"][hk - 1], g[f"dec2_hook{hk - 1}"])
    bad = {k: v for k, v in errs.items() if v > tol}
    assert not bad, bad
    if "tap_dpt_path1_0" in g:        # intermediate taps (tiny_32x32_b1): decoder input
        assert rel_l2(r["dec1"][0], g["dec1_in"]) < TOL


def test_post_sta_reductions_vs_reference_golden():
    """oracle restatement of estimate_intrinsic_from_pts3d / estimate_scale... vs vectors
    produced by the reference's own functions (incl. Z == 0 and conf == 0 pixels)."""
    g = load_golden("post")[0]
    assert max_rel(O.estimate_intrinsic_from_pts3d(g["pts"], g["conf"], True), g["K_shared"]) < 2e-6
    assert max_rel(O.estimate_intrinsic_from_pts3d(g["pts"], g["conf"], False), g["K_per"]) < 2e-6
    s = O.estimate_scale_with_depth_and_confidence(g["pts"][0, ..., 2], g["pts"][1, ..., 2], g["conf"][0], g["conf"][1])
    assert abs(float(s) - float(g["scale"])) < 2e-6 * abs(float(g["scale"]))


def test_oracle_world_pointcloud_and_se3_vs_reference_golden():
    """f4: oracle restatement of slam.py:396-408 vs the golden made with the reference's compute_local_pointclouds."""
    from helpers import load_golden, max_rel
    from oracle import sta_oracle as O
    g = load_golden("fmt")[0]
    pts, col = O.world_pointcloud(g["depths"], g["scales"], g["intrinsics"], g["poses"], g["confs"], g["imgs"], float(g["thres"]))
    assert pts.shape == g["points"].shape
    assert max_rel(pts, g["points"]) < 1e-5 and np.abs(col - g["colors"]).max() < 1e-6
    se3 = O.mat_to_se3(g["poses"])
    q = g["quat_xyzw"] * np.sign(g["quat_xyzw"][:, 3:4])
    assert np.abs(se3[:, 3:] - q).max() < 2e-6 and np.array_equal(se3[:, :3].astype(np.float32), g["poses"][:, :3, 3])


def _f2_case(name, cfg):
    g, meta = load_golden(name)
    H, W_, nview = int(meta["H"]), int(meta["W"]), int(meta["nview"])
    sd = W.state_dict(cfg, seed=int(meta["seed"]))
    imgs = W.synth_images(nview, H, W_, seed=int(meta["seed"]), tag=int(meta["tag"]))
    return g, meta, H, W_, nview, sd, imgs


@pytest.mark.parametrize("case", ["f2_tiny_48x64", "f2_tiny_80x48_portrait"])
def test_regress_two_views_f2_vs_reference_golden(case):
    """oracle.regress_two_views (restating slam.py:153-189) vs the golden produced by replaying that method
    on the reference model + the reference's slam_utils: accepted and rejected edges, the adjacent-edge exemption.
    The portrait case holds transposed maps [2,W,H] and the intrinsics the reference derives from them."""
    g, meta, H, W_, nview, sd, imgs = _f2_case(case, W.TINY)
    feats = [O.encode_image(W.TINY, sd, imgs[v:v + 1]) for v in range(nview)]
    i = nview - 1
    acc = g["accepted"]
    assert acc.any() and not acc.all() and acc[-1]          # fixture holds both outcomes; the adjacent edge is accepted
    for j in range(i):
        pose, c, confs, intri, depths = O.regress_two_views(W.TINY, sd, feats[i][0], feats[j][0], feats[i][1], feats[j][1],
                                                            i - j == 1, float(g["thres"]), H, W_)
        assert abs(c - float(g[f"conf_{j}"])) < 2e-6
        assert rel_l2(pose, g[f"pose_{j}"]) < TOL
        assert (confs is not None) == bool(acc[j]), (j, c, float(g["thres"]))
        if confs is None:
            assert intri is None and depths is None
            continue
        assert rel_l2(confs, g[f"confs_{j}"]) < TOL and rel_l2(depths, g[f"depths_{j}"]) < TOL
        assert max_rel(intri, g[f"intri_{j}"]) < TOL
    # the adjacent edge would be rejected by its confidence alone: the exemption of slam.py:169 is what accepts it
    assert float(g[f"conf_{i - 1}"]) < float(g["thres"])


def test_encode_image_normalize_true_vs_reference_golden():
    """_encode_image(normalize=True) (sta_model.py:172-173): enc_norm after the blocks."""
    g, meta, H, W_, nview, sd, imgs = _f2_case("f2_tiny_48x64", W.TINY)
    x, _pos = O.encode_image_normalized(W.TINY, sd, imgs[nview - 1:nview])
    assert rel_l2(x, g["enc_feat_norm"]) < TOL


@pytest.mark.parametrize("case", ["tiny_32x32_b1", "tiny_48x64_b2_sharp", "tiny_80x48_b2_portrait"])
def test_torch_cpu_port_vs_reference_golden(case):
    """oracle/torch_cpu.py (the torch-CPU re-expression bench.py times as the CPU baseline) vs the reference goldens."""
    from oracle import torch_cpu as T
    g, meta = load_golden(case)
    H, W_, B = int(meta["H"]), int(meta["W"]), int(meta["B"])
    sd = W.state_dict(W.TINY, seed=int(meta["seed"]), qk_gain=float(meta["qk_gain"]))
    imgs = W.synth_images(2 * B, H, W_, seed=int(meta["seed"]), tag=0)
    r = T.forward_pair(W.TINY, sd, imgs[:B], imgs[B:])
    tol = TOL_SHARP if float(meta["qk_gain"]) != 1.0 else TOL
    for side in ("main", "supp"):
        for k, gk in (("pts3d", "pts3d"), ("conf", "conf"), ("pose", "pose"), ("pose_conf", "pose_conf")):
            assert rel_l2(r[side][k], g[f"{side}_{gk}"]) < tol, (side, k)


@pytest.mark.parametrize("case", ["seq_tum_tiny_48x64", "seq_7scenes_tiny_48x64", "seq_tum_tiny_48x64_t075", "seq_default_tiny_48x64"])
def test_keyframe_sequence_oracle_vs_reference_golden(case):
    """The multi-keyframe replay of OnlineSLAM.step's frontend calls (oracle/gen_golden.py gen_seq: add_view + connect_view_i_j
    over a growing cache in the tumrgbd.yaml / 7scenes.yaml edge regimes, slam.py:142-241,244-297) through the ORACLE's
    restatements - encode_image, regress_two_views, estimate_scale_with_depth_and_confidence - against the reference golden:
    every edge's pose, confidence, decision, shared K, depth / confidence maps and the scale edges between overlapping edges."""
    from helpers import compare_seq_edges, seq_meta
    from oracle.seq_protocol import seq_edge_list, seq_frames
    g, meta = load_golden(case)
    m = seq_meta(meta)
    H, W_ = m["H"], m["W"]
    sd = W.state_dict(W.TINY, seed=m["seed"])
    frames = seq_frames(W, m["nkf"], H, W_, m["seed"], m["tag"])
    feats, first, edges = [], {}, []
    for i in range(m["nkf"]):
        feats.append(O.encode_image(W.TINY, sd, frames[i:i + 1]))
        js, _far = seq_edge_list(i, m["neighbor_edge_num"], m["loop_edge_num"], m["loop_dist_min"])
        for j in js:
            pose, c, confs, intri, depths = O.regress_two_views(W.TINY, sd, feats[i][0], feats[j][0], feats[i][1], feats[j][1],
                                                                i - j == 1, float(g["thres"]), H, W_)
            rec = dict(i=i, j=j, pose=pose, conf=c, accepted=confs is not None, confs=confs, intri=intri, depths=depths,
                       scales=[None, None], scale_confs=[None, None])
            if confs is not None:
                for k, v in enumerate((i, j)):                          # node bookkeeping (slam.py:203-218)
                    if v in first:
                        d0, c0 = first[v]
                        rec["scales"][k] = float(O.estimate_scale_with_depth_and_confidence(depths[k], d0, confs[k], c0))
                        rec["scale_confs"][k] = float(np.sqrt(confs[k] * c0).mean())
                    else:
                        first[v] = (depths[k], confs[k])
            edges.append(rec)
    acc = [e["accepted"] for e in edges]
    assert any(acc) and not all(acc)
    worst = compare_seq_edges(edges, g, meta, tol=5e-5)      # (scale edges: judged against their conditioning, helpers.compare_seq_edges)
    assert any(e["scales"][0] is not None or e["scales"][1] is not None for e in edges), "fixture holds no scale edge"
    assert worst["pose"] < TOL


@pytest.mark.parametrize("case", ["decpos_tiny_48x64_b2", "decpos_tiny_48x80_sharp"])
def test_decode_stereo_foreign_positions_oracle_vs_reference_golden(case):
    """_decode_stereo with positions that are not the patch grid (shifted windows, reversed order, another grid of the same token
    count): the reference rotates q / k by whatever it is handed (sta_blocks.py:134-137,196-199) - fixtures from the reference
    (oracle/gen_golden.py gen_decpos), the oracle's decode_stereo on the recorded encoder features and positions."""
    g, meta = load_golden(case)
    sd = W.state_dict(W.TINY, seed=int(meta["seed"]), qk_gain=float(meta["qk_gain"]))
    tol = TOL_SHARP if float(meta["qk_gain"]) != 1.0 else TOL
    tags = sorted({k.split("_pos_a")[0] for k in g if k.endswith("_pos_a")})
    assert tags == ["flip", "regrid", "shift"], tags
    last = W.TINY.hooks[-1] - 1
    for tag in tags:
        d1, d2 = O.decode_stereo(W.TINY, sd, g["enc_feat_a"], g["enc_feat_b"], g[f"{tag}_pos_a"], g[f"{tag}_pos_b"])
        errs = {}
        for hk in W.TINY.hooks[1:]:
            errs[f"dec1_hook{hk - 1}"] = rel_l2(d1[hk - 1], g[f"{tag}_dec1_hook{hk - 1}"])
            errs[f"dec2_hook{hk - 1}"] = rel_l2(d2[hk - 1], g[f"{tag}_dec2_hook{hk - 1}"])
        bad = {k: v for k, v in errs.items() if v > tol}
        assert not bad, (tag, bad)
        # the positions matter: the same features under the patch grid give a visibly different decoder output
        assert rel_l2(g["grid_dec1_last"], g[f"{tag}_dec1_hook{last}"]) > 30 * tol, tag
