"""CPU (build container only: needs /root/reference): the real-checkpoint acceptance kit of oracle/gen_golden.py end to end on a
stand-in checkpoint FILE - torch.save({'model': procedural tiny weights}) -> `--checkpoint` path -> fixtures - which must be
bit-equal to what the generator writes for the same weights handed over directly, carry the fingerprint of the file, and come
with the activation-range table.  (The full-architecture stand-in run is `python oracle/check_oracle_vs_ref.py ckpt`: minutes.)"""
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.skipif(not os.path.isdir("/root/reference"), reason="the reference tree exists in the build container only")


def test_checkpoint_mode_reproduces_the_procedural_fixtures(tmp_path, monkeypatch):
    import sys
    import torch
    sys.path.insert(0, ROOT)
    from oracle import gen_golden as GG
    from vista_slam_amd import weights as W
    monkeypatch.setattr(GG, "OUT", str(tmp_path))
    sd = W.state_dict(W.TINY, seed=43)
    ck = tmp_path / "tiny_standin.pth"
    torch.save({"model": {k: torch.from_numpy(v.copy()) for k, v in sd.items()}, "epoch": 3}, str(ck))
    # the kit, restricted to its 224x224 forward fixture (+ the range table) ...
    GG.gen_checkpoint(str(ck), tag="ckpttest", cfg=W.TINY, only={"ckpttest_224_b1"})
    # ... and the same case / a short sequence with the weights handed over directly
    GG.run_case(name="direct_224_b1", cfg=W.TINY, H=224, W_=224, B=1, sub=8)
    a, b = np.load(tmp_path / "ckpttest_224_b1.npz"), np.load(tmp_path / "direct_224_b1.npz")
    shared = [k for k in b.files if not k.startswith("meta_")]
    assert len(shared) > 20
    for k in shared:
        assert np.array_equal(a[k], b[k]), k
    assert str(a["ckpt_fingerprint"]) == W.state_dict_fingerprint(sd) and int(a["ckpt_tensors"]) == len(sd)
    assert a["range_enc_absmax"].shape == (W.TINY.enc_depth,) and a["range_dec_absmax"].shape == (W.TINY.dec_depth,)
    assert np.all(a["range_enc_absmax"] > 0) and a["range_dpt_act_absmax"].shape == (4,)
    table = (tmp_path / "ckpttest_ranges.txt").read_text()
    assert "encoder residual stream" in table and "LayerNorm gains" in table
    seq = dict(cfg=W.TINY, H=48, W_=64, nkf=4, neighbor_edge_num=3, loop_edge_num=2, rel_pose_thres=0.75, sub=4)
    GG.gen_seq(name="seq_ck", sd=GG.load_checkpoint_sd(str(ck)), extra={"ckpt_fingerprint": np.array("x")}, **seq)
    GG.gen_seq(name="seq_direct", **seq)
    a, b = np.load(tmp_path / "seq_ck.npz"), np.load(tmp_path / "seq_direct.npz")
    for k in [k for k in b.files if not k.startswith("meta_")]:
        assert np.array_equal(a[k], b[k], equal_nan=a[k].dtype.kind == "f"), k


def test_fingerprint_is_order_and_container_independent():
    import sys
    import torch
    sys.path.insert(0, ROOT)
    from vista_slam_amd import weights as W
    sd = W.state_dict(W.TINY, seed=43)
    fp = W.state_dict_fingerprint(sd)
    keys = list(sd)[::-1]
    assert W.state_dict_fingerprint({k: torch.from_numpy(sd[k].copy()) for k in keys}) == fp
    other = dict(sd); k0 = keys[0]; other[k0] = sd[k0] + np.float32(1e-3)
    assert W.state_dict_fingerprint(other) != fp
