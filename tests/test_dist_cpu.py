"""CPU, world_size=2, gloo: the multi-GPU layer of the STA path (pair sharding + all-gather of the
compact per-pair outputs).  The compute itself has no CPU path, so each rank fabricates deterministic
per-pair outputs; what is checked is that every rank reassembles ALL pairs in the global order."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from vista_slam_amd import parallel as P

H, W_ = 4, 6


def fake_outputs(pair_ids):
    """Deterministic stand-in for forward_pair outputs of the given global pair ids."""
    def one(side):
        B = len(pair_ids)
        ids = torch.tensor(pair_ids, dtype=torch.float32).view(B, 1, 1)
        return {"relative_pose": ids.view(B, 1, 1) * 10 + side + torch.arange(16.).view(1, 4, 4),
                "relative_pose_conf": ids.view(B) * 0.01 + side,
                "pts3d_pred": (ids.view(B, 1, 1, 1) + torch.arange(H * W_ * 3.).view(1, H, W_, 3) * 0.001 + side),
                "conf": ids.view(B, 1, 1) + 1 + torch.arange(H * W_ * 1.).view(1, H, W_) * 0.01 + side}
    return one(0), one(1)


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, num_pairs, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        lo, hi = P.shard_range(num_pairs, world, rank)
        main, supp = fake_outputs(list(range(lo, hi)))
        packed = P.pack_compact(main, supp)
        allp = P.gather_compact(packed, num_pairs)
        gm, gs = P.unpack_compact(allp, H, W_)
        rm, rs = fake_outputs(list(range(num_pairs)))
        ok = allp.shape == (num_pairs, P.compact_elems_per_pair(H, W_))
        for got, ref in ((gm, rm), (gs, rs)):
            ok &= torch.equal(got["relative_pose"], ref["relative_pose"])
            ok &= torch.equal(got["relative_pose_conf"], ref["relative_pose_conf"])
            ok &= torch.equal(got["depth"], ref["pts3d_pred"][..., 2])
            ok &= torch.equal(got["conf"], ref["conf"])
        ret[rank] = bool(ok)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("num_pairs", [16, 5, 2])   # equal shards (bench), ragged shards, one pair per rank
def test_two_rank_gather_reassembles_all_pairs(num_pairs):
    world = 2
    ret = mp.Manager().dict()
    mp.spawn(_worker, args=(world, _free_port(), num_pairs, ret), nprocs=world, join=True)
    assert dict(ret) == {0: True, 1: True}


def test_shard_range_covers_everything_once():
    for n in (0, 1, 5, 8, 16, 17):
        for world in (1, 2, 3, 8):
            spans = [P.shard_range(n, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


def test_single_process_gather_is_identity():
    main, supp = fake_outputs([0, 1, 2])
    p = P.pack_compact(main, supp)
    assert P.gather_compact(p, 3) is p


# ---------------------------------------------------------------------------------------------------------
# SLAM-side partitioning: the candidate edges of one keyframe scattered over the ranks, compact results all-gathered.
class _Edge:
    def __init__(self, e, accepted):
        g = torch.Generator().manual_seed(100 + e)
        self.accepted, self.rel_pose_conf = accepted, 0.1 * e + 0.05
        self.pose = torch.rand(4, 4, generator=g) + e
        self.intri = torch.rand(3, 3, generator=g) + e if accepted else None
        self.depths = torch.rand(2, H, W_, generator=g) + e if accepted else None
        self.confs = torch.rand(2, H, W_, generator=g) + 1 if accepted else None


def _edge_worker(rank, world, port, num_edges, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        accepted = [e % 3 != 1 for e in range(num_edges)]           # some loop candidates rejected
        mine = P.edge_shard(num_edges, world, rank)
        local = P.pack_edges([_Edge(e, accepted[e]) for e in mine], H, W_, device="cpu")
        allr = P.gather_edges(local, num_edges, H, W_)
        ok = len(allr) == num_edges
        for e, d in enumerate(allr):
            ref = _Edge(e, accepted[e])
            ok &= d["accepted"] == ref.accepted and abs(d["rel_pose_conf"] - ref.rel_pose_conf) < 1e-6
            ok &= torch.equal(d["pose"], ref.pose)
            if ref.accepted:
                ok &= torch.equal(d["intri"], ref.intri) and torch.equal(d["depths"], ref.depths) and torch.equal(d["confs"], ref.confs)
            else:
                ok &= d["intri"] is None and d["depths"] is None and d["confs"] is None
        ret[rank] = bool(ok)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("num_edges", [5, 6, 1])    # TUM-style keyframe (3 neighbours + 2 loops), even split, fewer edges than ranks
def test_two_rank_keyframe_edge_scatter(num_edges):
    world = 2
    assert sorted(P.edge_shard(num_edges, world, 0) + P.edge_shard(num_edges, world, 1)) == list(range(num_edges))
    ret = mp.Manager().dict()
    mp.spawn(_edge_worker, args=(world, _free_port(), num_edges, ret), nprocs=world, join=True)
    assert dict(ret) == {0: True, 1: True}


def test_pack_edges_portrait_views_round_trip_and_reject_mismatched_dims():
    """Portrait frames: regress_views returns TRANSPOSED views of [2, H_img, W_img] storage (slam_scheduler.py, like the
    reference's utils/misc.py:60-61,81).  pack_edges / gather_edges take the dims of the views; the image dims are refused
    instead of silently scrambling the maps."""
    Hi, Wi = 12, 8                                        # portrait image: H_img > W_img

    class PEdge(_Edge):
        def __init__(self, e):
            super().__init__(e, True)
            g = torch.Generator().manual_seed(7 + e)
            self.depths = torch.rand(2, Hi, Wi, generator=g).swapaxes(1, 2)     # [2, W_img, H_img] view, non-contiguous
            self.confs = torch.rand(2, Hi, Wi, generator=g).swapaxes(1, 2)
    edges = [PEdge(0), PEdge(1)]
    with pytest.raises(ValueError):
        P.pack_edges(edges, Hi, Wi, device="cpu")
    out = P.gather_edges(P.pack_edges(edges, Wi, Hi, device="cpu"), 2, Wi, Hi)
    for e, d in zip(edges, out):
        assert torch.equal(d["depths"], e.depths) and torch.equal(d["confs"], e.confs)


# ---------------------------------------------------------------------------------------------------------
# bench.py's own step / timing / reporting code under two ranks (the driver's --gpus N run is the first time it meets more
# than one RCCL rank: everything but the backend and the device is exercised here, on gloo + CPU tensors).
def _bench_worker(rank, world, port, ret):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    env = {"WORLD_SIZE": str(world), "RANK": str(rank), "LOCAL_RANK": str(rank), "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port)}
    os.environ.update(env)
    w, r, local, use_dist = bench.rank_env()
    assert (w, r, local, use_dist) == (world, rank, rank, True)
    dist.init_process_group("gloo", rank=r, world_size=w)
    try:
        B = 3
        ids = list(range(r * B, (r + 1) * B))                 # rank r owns pairs [r*B, (r+1)*B): weak scaling, B pairs per rank
        calls = [0]

        def forward():
            calls[0] += 1
            return fake_outputs(ids)
        runner = bench.StepRunner(forward, B, H, W_, w, use_dist, "cpu", model=None)
        runner.step()                                          # warm-up, untimed
        sums = runner.verify_gather()                          # first-run hardening: a non-empty record from EVERY rank after step 1
        assert len(sums) == world and all(v > 0 for v in sums)
        runner.step()
        steps = 5
        dt, step_ms, out = bench.timed_region(runner, steps)
        assert calls[0] == 2 + steps and len(step_ms) == steps and dt > 0       # EXACTLY K timed steps
        dt_max, fields = bench.distributed_fields(runner, dt, steps, sums)
        assert fields["world_size_reported_by_backend"] == world and fields["backend"] == "gloo"
        assert [i["rank"] for i in fields["ranks"]] == list(range(world)) and all(len(i["first_step_checksums"]) == world for i in fields["ranks"])
        assert len({i["pid"] for i in fields["ranks"]}) == world                     # one process per rank, each reported by itself
        ok = dt_max >= dt and len(fields["per_rank_pairs_per_s"]) == world and fields["all_gather_ms_median"] is not None
        ok &= fields["all_gather_bytes_per_rank"] == B * P.compact_elems_per_pair(H, W_) * 4
        # the last step's receive buffer holds every rank's records in global pair order
        gm, gs = P.unpack_compact(runner.last_gathered(), H, W_)
        rm, rs = fake_outputs(list(range(world * B)))
        for got, ref in ((gm, rm), (gs, rs)):
            ok &= torch.equal(got["relative_pose"], ref["relative_pose"]) and torch.equal(got["depth"], ref["pts3d_pred"][..., 2])
            ok &= torch.equal(got["conf"], ref["conf"]) and torch.equal(got["relative_pose_conf"], ref["relative_pose_conf"])
        ret[rank] = bool(ok)
    finally:
        dist.destroy_process_group()


def test_bench_step_runner_two_ranks_gloo():
    world = 2
    ret = mp.Manager().dict()
    mp.spawn(_bench_worker, args=(world, _free_port(), ret), nprocs=world, join=True)
    assert dict(ret) == {0: True, 1: True}


def test_verify_gather_catches_a_missing_shard():
    """A rank whose records never arrive (an all-zero slice of the receive buffer) fails the first-step check loudly."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    B = 2
    runner = bench.StepRunner(lambda: fake_outputs([0, 1]), B, H, W_, 2, False, "cpu", model=None)
    runner.use_dist, runner.n = True, 1
    full = P.pack_compact(*fake_outputs([0, 1, 2, 3]))
    runner.gathered = [full.clone(), full.clone()]
    assert len(runner.verify_gather()) == 2
    runner.gathered[0][B:] = 0.0                                                     # rank 1's shard missing
    with pytest.raises(AssertionError, match="empty"):
        runner.verify_gather()


def test_bench_rank_env_and_local_device(monkeypatch):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    assert bench.rank_env({}) == (1, 0, 0, False)                                   # bare `python bench.py`
    assert bench.rank_env({"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"}) == (1, 0, 0, True)    # one rank under a launcher: RCCL path
    assert bench.rank_env({"WORLD_SIZE": "8", "RANK": "5", "LOCAL_RANK": "5"}) == (8, 5, 5, True)
    with pytest.raises(AssertionError):
        bench.rank_env({"WORLD_SIZE": "2", "RANK": "2", "LOCAL_RANK": "0"})
    assert bench.local_device(3, 8) == "cuda:3"                                       # sta_create(device = LOCAL_RANK)
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "0")
    with pytest.raises(AssertionError, match="HIP_VISIBLE_DEVICES=0"):
        bench.local_device(1, 1)                                                      # a launcher narrowed the devices but kept LOCAL_RANK


def test_pack_compact_into_a_given_buffer_cpu():
    main, supp = fake_outputs([4, 7])
    want = P.pack_compact(main, supp)
    out = torch.full_like(want, -1.0)
    got = P.pack_compact(main, supp, model=None, out=out)
    assert got.data_ptr() == out.data_ptr() and torch.equal(out, want)
