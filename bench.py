#!/usr/bin/env python
"""bench.py - STA two-view image-pairs/sec on synthetic 512x384 RGB pairs (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One "step" = one pass of the whole hot path (2 x encode + decode + 2 x pose head + 2 x DPT head,
sta_model.py:247-291) over a batch of 8 synthetic image pairs per GPU, inputs resident in HBM.
Pairs are independent units: ranks process disjoint pairs (weak scaling); the only exchange is one
RCCL all-gather per step of the compact per-pair outputs a SLAM consumer reads (pose 4x4, pose
confidence, depth = pts[...,2] and the confidence map; slam.py:165-185).

`python bench.py --gpus N` with N > 1 from a bare shell (no WORLD_SIZE in the environment) starts its own N ranks
by re-executing itself under torch.distributed.run on 127.0.0.1.

Prints ONE JSON line on rank 0 with the throughput (`value` = K steps / wall time between two
barrier + synchronize brackets, max over ranks; `median_ms_per_step` = median of the K per-step stream-event
durations of rank 0), the roofline of the dominant kernel (live HIP event timing of every launch of the in-place
residual GEMM class inside the timed region - those per-launch event records are part of the timed region) and a CPU
baseline (a torch-CPU re-expression of the forward and the C + OpenMP oracle, timed on the host cores on a bounded
sample).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

H, W_, PAIRS_PER_GPU = 384, 512, 8
PEAK_F16_MFMA_TFLOPS = 2500.0      # dense fp16/bf16 MFMA peak, MI355X_MICROARCH.md "Chip-level parameters"
TIMED_EVERY = 4                    # inside the timed region every 4th launch of the dominant kernel symbol carries a HIP-event pair: a pair
                                   # costs ~9 us of dispatch (tools/probes/boundary_probe.hip: 11.4 vs 2.7 us per launch, profiles/
                                   # r05_boundary_probe.txt) - on all 36 launches per step that was 0.7 % of `value`


def pmc_traffic(kernel_prefix):
    """HBM bytes per launch of the dominant kernel from the newest committed rocprofv3 PMC summary
    (profiles/rNN_pmc_traffic.json <- tools/pmc_summary.py; FETCH_SIZE doubled per the gfx950
    correction).  bench.py cannot collect PMCs itself (they need their own rocprofv3 --pmc passes), so this is a
    STATIC figure from the profiled run of the same command, not a measurement of this run: the source file is
    reported beside it.  (None, None) when no summary is committed."""
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")), reverse=True):
        try:
            with open(path) as f:
                d = json.load(f)
            for k, v in d.items():
                if k.startswith(kernel_prefix):
                    return int(v["avg_hbm_bytes_per_launch"]), os.path.relpath(path, ROOT) + " (static: separate rocprofv3 --pmc passes)"
        except Exception:   # noqa: BLE001
            pass
    return None, None


def summarize_gemm_records(recs, precision):
    """Per kernel symbol totals of (M, N, K, epilogue, a_mode, mx, family, ms) launch records."""
    sp = "false" if precision == "f16" else "true"     # SPLIT template flag of the kernel symbol
    fam_tpl = {2: "256, 256, 4, 4, 0, 2", 3: "192, 256, 3, 4, 0, 2", 5: "192, 128, 2, 4, 0, 2", 6: "128, 64, 2, 2, 0, 3"}
    groups = {}
    for (M_, N_, K_, epi, amode, mx, fam, ms_) in recs:
        sym = (f"gemm_kernel<{sp}, {amode}, {epi}>" if fam == 1 else
               f"gemm2_pair_kernel<{sp}, {amode}, {epi}, 192, 128, 2, 4>" if fam == 7 else      # decoder: attn.qkv + cross_attn.projk|projv
               f"conv3h_kernel<{sp}, {epi}, 256, {128 if N_ == 128 else 256}, 4, 4, {'true' if mx else 'false'}>" if fam == 8 else   # halo-tiled 3x3 convolution
               f"gemm2_kernel<{sp}, {amode}, {epi}, {fam_tpl[fam]}, {'true' if mx else 'false'}>")
        g = groups.setdefault(sym, {"ms": 0.0, "fl": 0.0, "by": 0.0, "n": 0, "mx": mx, "epi": epi, "amode": amode, "fam": fam})
        g["ms"] += ms_; g["fl"] += 2.0 * M_ * N_ * K_; g["n"] += 1
        # algorithmic bytes: A and W planes (4 B per element in the split formats) once, output once (+ residual read)
        out_b = 16.0 * M_ if epi == 6 else 4.0 * M_ * N_ * (2.0 if epi == 5 else 1.0)     # fused DPT tail: pts (12 B) + conf (4 B) per pixel
        g["by"] += 4.0 * (float(M_) * K_ / (9.0 if amode else 1.0) + float(N_) * K_) + out_b
    return groups


def slam_probe(model, dev, iters=20):
    """Secondary figure: latency of the split entry points exactly as OnlineSLAM calls them
    (slam.py:144,162,165,179-180) at the SLAM resolution 224x224, batch 1 (launch-bound regime)."""
    import torch
    from vista_slam_amd import weights as Wt
    imgs = torch.from_numpy(Wt.synth_images(2, 224, 224, seed=43, tag=7)).to(dev)
    ts = torch.tensor([[224, 224]])

    def timed(fn):
        for _ in range(3):
            out = fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(iters):
            out = fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / iters * 1e3, out
    enc_ms, (fa, pa) = timed(lambda: model._encode_image(imgs[:1], ts, normalize=False))
    fb, pb = model._encode_image(imgs[1:], ts, normalize=False)
    dec_ms, (d1, d2) = timed(lambda: model._decode_stereo(fa, fb, pa, pb))
    pose_ms, _ = timed(lambda: model.head_pose_s(d1[-1][:, 0, :]))
    dpt_ms, _ = timed(lambda: model.head_pts([fa] + [t[:, 1:, :] for t in d1], ts))
    pair_ms = dec_ms + pose_ms + 2 * dpt_ms
    # f2: the same keyframe (5 accepted edges) through the batched native scheduler (sta_regress_views)
    from vista_slam_amd.slam_scheduler import regress_views
    sched_ms, _ = timed(lambda: regress_views(model, fa, [fb] * 5, [True] * 5, 0.0, 224, 224))
    return {"scheduler_5edges_ms": round(sched_ms, 3),
            "keyframes_per_s_scheduler": round(1e3 / (enc_ms + sched_ms), 2),
            "encode_ms": round(enc_ms, 3), "decode_ms": round(dec_ms, 3), "pose_ms": round(pose_ms, 3),
            "dpt_ms_per_view": round(dpt_ms, 3), "accepted_pair_ms": round(pair_ms, 3),
            "keyframes_per_s_est_5pairs": round(1e3 / (enc_ms + 5 * pair_ms), 2),
            "note": "frontend-only estimate for a TUM-style keyframe (1 encode + 5 accepted pairs); the reference's CPU stages (ORB/DBoW3, PGO) are not included"}


def slam_replay(model, dev, frames=120, warm=12, res=(224, 224), src_hw=(480, 640), neighbor_edge_num=3, loop_edge_num=2, streams=None):
    """Data-free stand-in for BASELINE configs[2] (TUM-RGBD through slam.py): `frames` synthetic 640x480 uint8 camera frames
    through the frontend's calls in `OnlineSLAM.step`'s order (slam.py:244-297) with a GROWING feature cache -
    f3 input step (crop / LANCZOS / ImgNorm, slam_images_only.py:19-33) -> add_view = encode (slam.py:142-151) -> the
    <= neighbor_edge_num neighbour edges and the <= loop_edge_num loop candidates (older views, chosen by a hash; the
    reference's DBoW3 detector is a CPU stage, out of scope, and reads only the grey image - slam.py:267 - so its candidates
    do not depend on the neighbour edges' results) as ONE batched scheduler call, edge order = the reference's
    (slam.py:263-277; regress_two_views + early reject, slam.py:153-189) -> per accepted edge the node bookkeeping's device
    work: the scale edge to the view's first node (estimate_scale_with_depth_and_confidence + the sqrt-mean confidence,
    slam.py:205-218) -> at the end the f4 world point cloud over every view (save_data_all, slam.py:396-408).  The rejection
    threshold is the 40 % quantile of the non-adjacent pose confidences of the warm-up frames, so ~40 % of the non-adjacent
    edges skip the DPT heads like rejected loop closures do.  No dataset, checkpoint, pypose, DBoW3: ATE cannot be produced
    here (said in DESIGN.md); what is measured is keyframes/s of the frontend + its f1-f4 neighbours and the per-stage split.

    Two schedules of the SAME calls: `single_stream` (everything in order on one stream) and the pipelined one reported as
    `keyframes_per_s`: three streams - f3 + encode of keyframe i+1; decode + pose heads of keyframe i's edges
    (regress_views_begin); DPT heads + reductions + node bookkeeping of keyframe i-1's accepted edges (regress_views_finish) -
    which only uses independence the SLAM loop itself has: add_view(i+1) needs no result of keyframe i (slam.py:258), and the
    edges of keyframe i need encoder features only (slam.py:153-162).  At 224x224, batch 1 every chain is ~200 dependent
    dispatches that each leave most of the chip idle; the library keeps one scratch context per stream."""
    import torch
    from vista_slam_amd import weights as Wt
    from vista_slam_amd.preprocess import process_image
    from vista_slam_amd.formats import world_pointcloud
    Hs, Ws = src_hw
    Wr, Hr = res
    n_all = frames + warm
    distinct = [torch.from_numpy(Wt.synth_frames_u8(Hs, Ws, seed=43, tag=t)).to(dev) for t in range(16)]
    raw = [distinct[f % 16] for f in range(n_all)]                      # frames resident in HBM (16 distinct ones, cycled)
    torch.cuda.synchronize()
    # the three lanes run on streams the LIBRARY hands out after measuring that they overlap pairwise (sta_pipeline_streams; round 4
    # timed its warm-up on four triples of torch streams and kept the best: 188 vs 231 keyframes/s depending on which streams share
    # a hardware queue).  `streams` overrides them (tools/queue_probe.py).
    placement = {"source": "caller"}
    if streams is None:
        streams = model.pipeline_streams(3)
        placement = {"source": "sta_pipeline_streams (library-owned, pairwise overlap measured by a spin probe)",
                     "verified_concurrent": model.pipeline_streams_verified}
    enc_stream, edge_streams = streams[0], list(streams[1:3])
    # everything the replay's calls will need on its streams, allocated NOW (sta_reserve): one 224x224 frame, up to
    # neighbor_edge_num + loop_edge_num candidate edges per keyframe, the current stream (single-stream schedule) and the three lanes.
    # (The f3 input step and the f4 cloud size themselves lazily: tables per source geometry / workspace per view count.)
    model.reserve(1, Hr, Wr, max_edges=neighbor_edge_num + loop_edge_num, streams=[torch.cuda.current_stream(dev)] + list(streams[:3]))

    def run(nf, thres, pipelined):
        from vista_slam_amd.keyframe_pipeline import replay
        rgbs = []
        poses = {0: torch.eye(4, device=dev)}
        ev = {k: [] for k in ("f3", "encode", "edges_decode", "edges_heads", "bookkeeping")}
        stats = {"edges": 0, "rejected": 0, "nonadj_conf": []}

        def mark():
            e = torch.cuda.Event(enable_timing=True); e.record(); return e

        def add_view(i):                        # f3 + encode of frame i on the CURRENT stream (slam.py:142-151)
            t0 = mark()
            pre = process_image(model, raw[i], resolution=(Wr, Hr))
            t1 = mark()
            feat, pos = model.encode_u8hwc(pre["u8"][None])
            t2 = mark()
            ev["f3"].append((t0, t1)); ev["encode"].append((t1, t2))
            rgbs.append(pre["rgb"])
            return feat, pos

        def edge_list(i):                       # neighbours (slam.py:262-265), then loop candidates among the older views (:273-277)
            far = max(0, i - neighbor_edge_num)
            js = list(range(far, i))
            if far >= 8:
                js += sorted({(i * 7919 + 13) % far, (i * 104729 + 7) % far})[:loop_edge_num]
            return js

        def on_edges(i, js, recs):              # host-side statistics + the pose chain of the harness (first accepted edge to a posed view)
            for r in recs:
                stats["edges"] += 1
                if r.i - r.j != 1:
                    stats["nonadj_conf"].append(r.rel_pose_conf)
                if not r.accepted:
                    stats["rejected"] += 1
                elif r.i not in poses and r.j in poses:
                    poses[r.i] = poses[r.j] @ r.pose

        _recs, book, _feats = replay(model, nf, add_view, edge_list, thres, Hr, Wr, schedule="pipelined" if pipelined else "batched",
                                     streams=[enc_stream] + edge_streams, timeline=ev, on_edges=on_edges, keep_records=False)
        stats["scale_edges"] = book.scale_edges
        first = book.first
        # f4: world point cloud of every view that has a node (slam.py:396-408)
        t4 = mark()
        ids = sorted(v for v in first if v in poses)
        npts = 0
        if ids:
            depths = torch.stack([first[v][0] for v in ids]); confs = torch.stack([first[v][1] for v in ids])
            Ks = torch.stack([first[v][2] for v in ids]); Ps = torch.stack([poses[v] for v in ids])
            imgs = torch.stack([rgbs[v] for v in ids])
            pts, _ = world_pointcloud(model, depths, torch.ones(len(ids), 1, device=dev), Ks, Ps, confs, imgs, float(confs.median()))
            npts = int(pts.shape[0])
        t5 = mark()
        torch.cuda.synchronize()
        ms = {k: sum(a.elapsed_time(b) for a, b in v) for k, v in ev.items()}
        ms["f1"] = ms.pop("bookkeeping")
        ms["f4"] = t4.elapsed_time(t5)
        return ms, stats, npts, len(ids)

    _, st_w, _, _ = run(warm, -1.0, False)                               # warm-up: f3 tables, threshold (the workspaces are reserved)
    run(warm, -1.0, True)
    alloc_after_warmup = model.alloc_stats()
    conf = sorted(st_w["nonadj_conf"])
    thres = conf[int(0.4 * len(conf))] if conf else -1.0
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ms_s, st_s, npts_s, _ = run(frames, thres, False)
    dt_s = time.perf_counter() - t0
    # the pipelined schedule twice: the first full-length pass still grows per-stream pools (torch's caching allocator keeps one
    # pool per stream, the library one workspace per stream - the 12-frame warm-up reaches neither high-water mark); the second
    # pass is the steady state a long-running SLAM process sees and is the one reported; the first is kept beside it
    dts = []
    for _rep in range(2):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ms, st, npts, nviews = run(frames, thres, True)
        dts.append(time.perf_counter() - t0)
    dt = dts[-1]
    alloc_end = model.alloc_stats()
    nonadj = len(st["nonadj_conf"])
    return {"frames": frames, "keyframes_per_s": round(frames / dt, 2), "ms_per_keyframe": round(dt / frames * 1e3, 3),
            "first_pass_keyframes_per_s": round(frames / dts[0], 2), "stream_placement": placement,
            "library_allocations_and_device_syncs_during_the_three_measured_passes": [alloc_end[0] - alloc_after_warmup[0], alloc_end[1] - alloc_after_warmup[1]],
            "same_result_as_single_stream": bool(npts == npts_s and st["rejected"] == st_s["rejected"] and st["edges"] == st_s["edges"]),
            "schedule": "three streams: f3 + encode of keyframe i+1 | decode + pose heads of keyframe i's edges (regress_views_begin) | DPT heads, "
                        "reductions and node bookkeeping of keyframe i-1 (regress_views_finish); one library scratch context per stream",
            "stage_ms_per_keyframe": {k: round(v / frames, 3) for k, v in ms.items()},
            "stage_note": "per-stage event intervals on each stage's own stream; under the pipelined schedule they overlap and do not add up",
            "single_stream": {"keyframes_per_s": round(frames / dt_s, 2), "ms_per_keyframe": round(dt_s / frames * 1e3, 3),
                              "stage_ms_per_keyframe": {k: round(v / frames, 3) for k, v in ms_s.items()}},
            "edges_per_keyframe": round(st["edges"] / frames, 2), "rejected_frac_of_non_adjacent": round(st["rejected"] / max(1, nonadj), 3),
            "scale_edges": st["scale_edges"], "views_in_cloud": nviews, "cloud_points": npts, "source_frames": f"{Ws}x{Hs} uint8 -> {Wr}x{Hr}",
            "note": "frontend + f1-f4 rows in OnlineSLAM.step order on synthetic frames with a growing feature cache (every frame a "
                    "keyframe); the reference's CPU stages (optical-flow keyframing, ORB / DBoW3 loop detection, pypose PGO) are not part of "
                    "it and no ATE can be produced without the dataset and the checkpoint"}


def rank_env(environ=None):
    """(world, rank, local_rank, use_dist) from the launcher's environment (torch.distributed.run sets WORLD_SIZE / RANK /
    LOCAL_RANK / MASTER_*).  use_dist is True whenever a launcher started us - also with ONE rank, which exercises the RCCL
    path on a 1-GPU box.  One process per GPU: rank r of a node drives device index LOCAL_RANK of the devices visible to it
    (HIP_VISIBLE_DEVICES, when a launcher narrows it per process, already renumbers them from 0 - then LOCAL_RANK must be 0)."""
    env = os.environ if environ is None else environ
    world = int(env.get("WORLD_SIZE", "1"))
    rank = int(env.get("RANK", "0"))
    local = int(env.get("LOCAL_RANK", "0"))
    assert 0 <= rank < world, f"RANK={rank} outside WORLD_SIZE={world}"
    return world, rank, local, "WORLD_SIZE" in env


def local_device(local, visible_count):
    """cuda:<LOCAL_RANK>, or a loud failure when the launcher handed this rank a GPU index it cannot see."""
    assert visible_count > local, (f"LOCAL_RANK={local} but only {visible_count} GPU(s) visible to this process "
                                   f"(HIP_VISIBLE_DEVICES={os.environ.get('HIP_VISIBLE_DEVICES', '<unset>')})")
    return f"cuda:{local}"


class StepRunner:
    """One bench step = `forward()` over this rank's B pairs + (under a launcher) the step's one exchange: the compact
    per-pair records (parallel.pack_compact: ONE kernel, written straight into the send buffer) all-gathered to every rank.
    On a GPU the gather of step i runs on its own stream into one of two receive buffers and overlaps the forward of step
    i+1; the closing device synchronize of the timed region waits for the last one, so every gather is inside the measured
    time.  The same code runs on CPU tensors with the gloo backend (tests/test_dist_cpu.py: world 2, fabricated outputs)."""

    def __init__(self, forward, B, H, W, world, use_dist, dev, model=None):
        import torch
        from vista_slam_amd import parallel as P
        self.torch, self.P = torch, P
        self.forward, self.B, self.H, self.W, self.world, self.use_dist, self.dev, self.model = forward, B, H, W, world, use_dist, dev, model
        self.cuda = torch.device(dev).type == "cuda"
        self.n = 0
        self.gather_ev, self.gather_ms_cpu = [], []
        self.send = self.gathered = self.comm_stream = None
        self.sent = [None, None]
        if use_dist:
            width = P.compact_elems_per_pair(H, W)
            self.send = [torch.empty(B, width, device=dev) for _ in range(2)]
            self.gathered = [torch.empty(world * B, width, device=dev) for _ in range(2)]
            if self.cuda:
                self.comm_stream = torch.cuda.Stream(device=dev)

    def step(self, timed=False):
        torch, P = self.torch, self.P
        main_o, supp_o = self.forward()
        if self.use_dist:
            k = self.n & 1
            if self.cuda:
                if self.sent[k] is not None:                 # the gather of step n-2 read this send buffer: it must be done
                    torch.cuda.current_stream(self.dev).wait_event(self.sent[k])
                P.pack_compact(main_o, supp_o, model=self.model, out=self.send[k])
                ready = torch.cuda.Event(); ready.record()
                with torch.cuda.stream(self.comm_stream):
                    self.comm_stream.wait_event(ready)
                    if timed:
                        e0 = torch.cuda.Event(enable_timing=True); e0.record()
                    P.gather_compact(self.send[k], self.world * self.B, out=self.gathered[k])
                    if timed:
                        e1 = torch.cuda.Event(enable_timing=True); e1.record()
                        self.gather_ev.append((e0, e1))
                    self.sent[k] = torch.cuda.Event(); self.sent[k].record()
            else:
                P.pack_compact(main_o, supp_o, model=None, out=self.send[k])
                t0 = time.perf_counter()
                P.gather_compact(self.send[k], self.world * self.B, out=self.gathered[k])
                if timed:
                    self.gather_ms_cpu.append((time.perf_counter() - t0) * 1e3)
            self.n += 1
        return main_o, supp_o

    def last_gathered(self):
        return None if not self.use_dist or self.n == 0 else self.gathered[(self.n - 1) & 1]

    def verify_gather(self):
        """After a step: on THIS rank, the last receive buffer must hold a non-trivial record from EVERY rank (a pose row has
        pose[3][3] = 1, so an all-zero slice means that rank's shard never arrived).  Returns the per-source-rank checksums;
        raises on a missing shard.  First-run hardening for N > 1: proves what the collective delivered, on every rank."""
        torch = self.torch
        g = self.last_gathered()
        assert g is not None, "verify_gather before the first distributed step"
        if self.cuda:
            torch.cuda.synchronize()
        sums = g.view(self.world, self.B, -1).abs().double().sum(dim=(1, 2)).cpu().tolist()
        finite = bool(torch.isfinite(g).all())
        assert finite and all(s_ > 0.0 for s_ in sums), f"all-gather delivered empty / non-finite shards: per-source-rank checksums {sums}"
        return sums

    def gather_ms(self):
        if self.cuda:
            return sorted(a.elapsed_time(b) for a, b in self.gather_ev)
        return sorted(self.gather_ms_cpu)


def timed_region(runner, steps, warmup_done=True):
    """EXACTLY `steps` steps between two (barrier + device synchronize) brackets -> (wall seconds of this rank, sorted per-step
    ms from stream events on a GPU / host clocks on CPU, last outputs)."""
    import torch
    import torch.distributed as dist
    cuda = runner.cuda

    def fence():
        if cuda:
            torch.cuda.synchronize()
        if runner.use_dist:
            dist.barrier()
    fence()
    marks = []

    def mark():
        if cuda:
            e = torch.cuda.Event(enable_timing=True); e.record(); marks.append(e)
        else:
            marks.append(time.perf_counter())
    t0 = time.perf_counter()
    mark()
    out = None
    for _ in range(steps):
        out = runner.step(timed=True)
        mark()                          # stream-ordered marker, no host sync inside the timed region
    fence()                             # all streams of the device, the communication stream included
    dt = time.perf_counter() - t0
    if cuda:
        step_ms = sorted(marks[i].elapsed_time(marks[i + 1]) for i in range(steps))
    else:
        step_ms = sorted((marks[i + 1] - marks[i]) * 1e3 for i in range(steps))
    return dt, step_ms, out


def rank_identity(dev):
    """What this rank actually drives: device name + PCI address (a launcher that maps two ranks onto one GPU shows up here)."""
    import socket
    import torch
    d = torch.device(dev)
    if d.type != "cuda":
        return {"host": socket.gethostname(), "device": "cpu", "pci": None, "pid": os.getpid()}
    p = torch.cuda.get_device_properties(d)
    pci = "%04x:%02x:%02x" % (getattr(p, "pci_domain_id", 0), getattr(p, "pci_bus_id", 0), getattr(p, "pci_device_id", 0))
    return {"host": socket.gethostname(), "device": p.name, "pci": pci, "index": d.index, "pid": os.getpid(),
            "hip_visible_devices": os.environ.get("HIP_VISIBLE_DEVICES")}


def distributed_fields(runner, dt, steps, first_step_checksums=None):
    """MAX over ranks of the timed wall clock + the per-rank rates, the gather statistics and - first-run hardening - what the
    process group really is: the world size the backend reports, every rank's device / PCI address (gathered over the group) and
    every rank's view of the first step's gathered records (per-source-rank checksums, `StepRunner.verify_gather`)."""
    import torch
    import torch.distributed as dist
    t = torch.tensor([dt], device=runner.dev, dtype=torch.float64)
    allt = [torch.zeros_like(t) for _ in range(runner.world)]
    dist.all_gather(allt, t)
    g = runner.gather_ms()
    ident = [None] * dist.get_world_size()
    dist.all_gather_object(ident, {"rank": dist.get_rank(), **rank_identity(runner.dev), "first_step_checksums": first_step_checksums})
    pcis = [i["pci"] for i in ident if i.get("pci")]
    assert dist.get_world_size() == runner.world, f"backend reports world {dist.get_world_size()}, launcher said {runner.world}"
    distinct = len(set((i["host"], i["pci"]) for i in ident if i.get("pci"))) == len(pcis)
    if not distinct:         # reported, not asserted: a driver that exposes no PCI address must not cost the run its line
        print(f"[bench] WARNING: ranks report the same (host, PCI address) - two ranks on one GPU? {ident}", file=sys.stderr, flush=True)
    return max(float(x.item()) for x in allt), {
        "per_rank_pairs_per_s": [round(runner.B * steps / float(x.item()), 3) for x in allt],
        "all_gather_ms_median": round(g[len(g) // 2], 4) if g else None,
        "all_gather_bytes_per_rank": int(runner.B * runner.P.compact_elems_per_pair(runner.H, runner.W) * 4),
        "world_size_reported_by_backend": dist.get_world_size(), "backend": dist.get_backend(),
        "distinct_gpus": distinct, "ranks": ident, "collective_library": collective_library()}


def collective_library():
    """Which library the ranks met through: torch's NCCL binding is RCCL on ROCm (its version as torch reports it; also echoed on
    stderr).  NCCL_DEBUG=VERSION would print the same banner - to STDOUT, in front of the one JSON line - so it is not set here."""
    import torch
    import torch.distributed as dist
    if dist.get_backend() != "nccl":
        return dist.get_backend()
    try:
        v = "RCCL " + ".".join(str(x) for x in torch.cuda.nccl.version()) + f" (torch {torch.__version__}, HIP {torch.version.hip})"
    except Exception as e:   # noqa: BLE001
        v = f"nccl backend (version query failed: {e})"
    print(f"[bench] collective library: {v}", file=sys.stderr, flush=True)
    return v


def host_cores():
    """Cores this process may actually use: min(cpu_count, affinity mask, cgroup CPU quota).  The GPU boxes report 256
    logical CPUs but run the container under a 16-CPU cgroup quota (cpu.max = 1600000 100000): 256 threads on that
    quota made the round-1 baseline 5x slower than 16 threads, and torch's CPU kernels stall outright."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except AttributeError:
        pass
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            with open(path) as f:
                tok = f.read().split()
            if path.endswith("cpu.max"):
                if tok[0] != "max":
                    n = min(n, max(1, int(int(tok[0]) / int(tok[1]))))
            else:
                q = int(tok[0])
                if q > 0:
                    with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f2:
                        n = min(n, max(1, int(q / int(f2.read().split()[0]))))
            break
        except (OSError, ValueError, IndexError):
            continue
    return n


def cpu_baseline_subprocess(H, W_, timeout_s=300):
    """Run cpu_baseline() in its own process under a hard timeout: a CPU-side stall must never cost the bench line."""
    import subprocess
    cores = host_cores()
    env = dict(os.environ, OMP_NUM_THREADS=str(cores), MKL_NUM_THREADS=str(cores), HIP_VISIBLE_DEVICES="")
    try:
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-baseline-only", str(H), str(W_)],
                           env=env, capture_output=True, text=True, timeout=timeout_s)
        for line in reversed(r.stdout.splitlines()):
            if line.startswith("{"):
                return json.loads(line)
        return {"error": f"no result (rc {r.returncode}): {r.stderr[-300:]}"}
    except subprocess.TimeoutExpired:
        return {"error": f"CPU baseline exceeded {timeout_s} s on {cores} cores and was stopped"}


def cpu_baseline(H=384, W_=512, with_c_port=True):
    """CPU baseline on the host cores, ONE pair of the metric's own workload (512x384, 1857.5 GFLOP) per timed pass:
    (1) `value`: oracle/torch_cpu.py, a torch-CPU (MKL / oneDNN) re-expression of the reference forward with
    torch.set_num_threads(all cores) - the kernels the reference's own CPU path would run on, best of 2 passes after one
    warm-up; (2) `c_port`: the C + OpenMP oracle (oracle/sta_oracle.py), the checker of the parity tests, one pass.
    Both are ports (the reference's Python does not travel to the GPU box); bounded to ~10-40 s of CPU work in total."""
    import torch
    from oracle import torch_cpu as T
    from vista_slam_amd import weights as Wt
    cores = host_cores()
    sd = Wt.state_dict(Wt.FULL, seed=43)
    imgs = Wt.synth_images(2, H, W_, seed=43, tag=0)
    gf = 1857.47 if (H, W_) == (384, 512) else 435.81
    torch.set_num_threads(cores)
    T.forward_pair(Wt.FULL, sd, imgs[:1], imgs[1:])                      # warm-up (thread pool, oneDNN primitive cache)
    best = 1e30
    for _ in range(2):
        t0 = time.perf_counter()
        T.forward_pair(Wt.FULL, sd, imgs[:1], imgs[1:])
        best = min(best, time.perf_counter() - t0)
    res = {"value": round(1.0 / best, 5), "unit": "pairs/s", "cores": cores, "kind": "port", "impl": "torch-cpu",
           "sample": f"1 pair @{W_}x{H} ({gf:.1f} GFLOP) x 3 passes, best of the last 2 = {best:.2f} s; oracle/torch_cpu.py "
                     f"(torch {torch.__version__} CPU kernels, torch.set_num_threads({cores}) = usable cores: "
                     f"{os.cpu_count()} logical CPUs under the container's cgroup CPU quota)",
           "gflops": round(gf / best, 1)}
    if with_c_port:
        from oracle import sta_oracle as O
        os.environ["OMP_NUM_THREADS"] = str(cores)
        t0 = time.perf_counter()
        O.forward_pair(Wt.FULL, sd, imgs[:1], imgs[1:])
        dt = time.perf_counter() - t0
        res["c_port"] = {"value": round(1.0 / dt, 5), "unit": "pairs/s", "cores": cores, "gflops": round(gf / dt, 1),
                         "sample": f"1 pair @{W_}x{H}, {dt:.1f} s, oracle/sta_oracle.py (C + OpenMP restatement; the parity checker)"}
    return res


def main():
    if len(sys.argv) >= 4 and sys.argv[1] == "--cpu-baseline-only":       # child of cpu_baseline_subprocess
        print(json.dumps(cpu_baseline(int(sys.argv[2]), int(sys.argv[3]))), flush=True)
        return
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--precision", default="f16x3h", choices=["f16x3h", "f16x3", "f16x3m"],
                    help="f16x3h = the default policy; f16x3m (opt-in, +2 %%): additionally mlp.fc2 in the f16mx arithmetic - holds every full-architecture "
                         "golden at <= 8.1e-5 but not two of the ten tiny stress sets (DESIGN.md section 2)")
    ap.add_argument("--pairs", type=int, default=PAIRS_PER_GPU, help="image pairs per GPU per step")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-baseline-224", action="store_true", help="time the oracle on one 224x224 pair (4.3x less work) instead of 512x384")
    ap.add_argument("--no-kernel-timing", action="store_true")
    ap.add_argument("--timed-every", type=int, default=TIMED_EVERY, help="the dominant kernel's launches inside the timed region: every N-th carries a HIP-event pair")
    ap.add_argument("--no-slam-probe", action="store_true")
    ap.add_argument("--slam-frames", type=int, default=120, help="frames of the slam_replay section (0 = skip)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # bare `python bench.py --gpus N`: start the N ranks ourselves (one process per GPU, RCCL) and relay the JSON line
        import socket
        import subprocess
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        sys.exit(subprocess.call(cmd, env=env))

    # stdout carries exactly ONE line, the JSON result: native libraries write there too (RCCL prints a five-line version banner
    # to stdout when its first communicator is created), so fd 1 is pointed at stderr for the whole run and the result line goes to
    # the saved descriptor at the end
    sys.stdout.flush()
    result_fd = os.dup(1)
    os.dup2(2, 1)

    import torch
    import torch.distributed as dist
    from vista_slam_amd import weights as Wt
    from vista_slam_amd.sta_frontend import STAFrontend

    world, rank, local, use_dist = rank_env()
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device(f"cuda:{local}"))
    assert args.gpus == world, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    dev = local_device(local, torch.cuda.device_count())
    torch.cuda.set_device(local)

    model = STAFrontend(Wt.FULL, dev, precision=args.precision).load_procedural(seed=43)      # sta_create(device = LOCAL_RANK)
    B = args.pairs
    model.reserve(B, H, W_)          # workspace, scratch context and side lane of this stream up front: no call below allocates (sta_reserve)
    alloc_reserved = model.alloc_stats()
    imgs = Wt.synth_images(2 * B, H, W_, seed=43, tag=rank)          # different pairs on every rank
    img_a = torch.from_numpy(imgs[:B]).to(dev)
    img_b = torch.from_numpy(imgs[B:]).to(dev)

    from vista_slam_amd import parallel as P
    runner = StepRunner(lambda: model.forward_pair(img_a, img_b), B, H, W_, world, use_dist, dev, model=model)
    step = runner.step

    first_sums = None
    for w in range(args.warmup):
        step()
        if w == 0 and use_dist:
            first_sums = runner.verify_gather()          # every rank: a non-empty record from every rank after the FIRST step
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    symtab = None
    if not args.no_kernel_timing:
        # untimed survey step: every GEMM / convolution launch carries an event pair (mode 2) -> per-symbol totals; the
        # symbol with the largest total time is the dominant kernel, and ONLY its launches are timed inside the timed region
        # (mode 3: a 1-in-TIMED_EVERY sample of them; an event pair costs ~9 us of dispatch)
        model.kernel_timing(2)
        step()
        torch.cuda.synchronize()
        symtab = summarize_gemm_records(model.kernel_timing_records(), args.precision)
        dom = max(symtab.values(), key=lambda g: g["ms"])
        model.kernel_timing(False)
        from vista_slam_amd import _lib
        _lib.check(model.lib.sta_kernel_timing_filter(model._h, dom["epi"], dom["amode"], dom["fam"], dom["mx"], args.timed_every))
        model.kernel_timing(3)
    alloc_before = model.alloc_stats()
    dt, step_ms, out = timed_region(runner, args.steps)
    alloc_after = model.alloc_stats()
    median_ms = step_ms[len(step_ms) // 2] if len(step_ms) % 2 else 0.5 * (step_ms[len(step_ms) // 2 - 1] + step_ms[len(step_ms) // 2])
    dist_fields = None
    if use_dist:
        if first_sums is None:
            first_sums = runner.verify_gather()
        dt, dist_fields = distributed_fields(runner, dt, args.steps, first_sums)
    assert bool(torch.isfinite(out[0]["pts3d_pred"]).all()), "non-finite output"

    roof = None
    if not args.no_kernel_timing:
        # Every launch of the dominant kernel SYMBOL (largest total time in the survey step; the rocprofv3 row of the same name
        # is directly comparable) carried a HIP-event pair in the timed region, recorded by the library on its launch stream;
        # achieved = sum of algorithmic 2MNK over those launches / sum of their event durations.
        epi_name = {0: "fp32 epilogue", 1: "fp16-plane epilogue", 2: "QKV + RoPE epilogue (attn.qkv, cross_attn.projq/k/v)",
                    3: "ConvTranspose scatter epilogue", 4: "GELU epilogue (mlp.fc1)", 5: "in-place residual epilogue (attn.proj, mlp.fc2, cross_attn.proj)",
                    6: "fused DPT tail (head.2 + ReLU + head.4 + point-map / confidence activations)"}
        groups = summarize_gemm_records(model.kernel_timing_records(), args.precision)
        model.kernel_timing(False)
        if groups:
            sym, g = max(groups.items(), key=lambda kv: kv[1]["ms"])
            ach = g["fl"] / (g["ms"] * 1e-3) / 1e12
            prod = 2 if g["mx"] else 3
            traffic, traffic_src = pmc_traffic(sym)
            tot_ms = sum(x["ms"] for x in symtab.values())          # survey step: all symbols, one step
            roof = {"bound": "mfma", "kernel": sym + " - " + ("3x3 convolution, " if g["amode"] else "") + epi_name[g["epi"]],
                    "achieved": round(ach, 1), "peak": PEAK_F16_MFMA_TFLOPS, "unit": "TFLOP/s",
                    "frac": round(ach / PEAK_F16_MFMA_TFLOPS, 4), "traffic": traffic, "traffic_source": traffic_src,
                    "timing": f"HIP events recorded by the library around every {args.timed_every}th launch of this kernel on its launch stream, inside "
                              "the timed region (their cost is part of `value`); `launches` = the timed ones",
                    "algorithmic_bytes_per_launch": int(g["by"] / g["n"]), "gflop_per_launch": round(g["fl"] / g["n"] / 1e9, 2),
                    "launches": g["n"], "avg_launch_us": round(g["ms"] * 1e3 / g["n"], 2),
                    "share_of_gemm_time": round(symtab[sym]["ms"] / tot_ms, 4),
                    "mfma_products_per_flop": prod, "issued_frac": round(ach * prod / PEAK_F16_MFMA_TFLOPS, 4),
                    "all_gemm_conv_kernels_survey_step": {
                        "achieved": round(sum(x["fl"] for x in symtab.values()) / (tot_ms * 1e-3) / 1e12, 1), "ms": round(tot_ms, 3),
                        "launches": sum(x["n"] for x in symtab.values()),
                        "note": "one untimed step with an event pair around every GEMM / convolution launch (slower than a timed-region step)"}}

    if rank == 0:
        pairs = B * world * args.steps
        flops_pair = model.flops_per_pair(H, W_)
        res = {"metric": "STA image-pairs/sec @512x384", "value": round(pairs / dt, 3), "unit": "pairs/s",
               "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
               "ms_per_step": round(dt / args.steps * 1e3, 3), "median_ms_per_step": round(median_ms, 3),
               "value_at_median_step": round(B * world / (median_ms * 1e-3), 3),
               "higher_is_better": True, "scaling": "weak",
               "vs_baseline": None, "dtype": {"f16x3": "f16x3-split MFMA (3 fp16 products per contraction), fp32 accumulate",
                                             "f16x3h": "f16x3-split MFMA (3 fp16 products) in the transformer and the pose head; fp16 MFMA + one block-scaled fp8 correction MFMA in the DPT head's convolutions; fp32 accumulate",
                                             "f16x3m": "OPT-IN policy, not the default: f16x3h + mlp.fc2 of both transformers as fp16 MFMA + one block-scaled fp8 correction MFMA; fp32 accumulate"}[args.precision],
               "data": "synthetic (uint8-uniform RGB pairs, procedural weights of the full 438M-parameter architecture)",
               "config": {"workload": f"512x384 batch={B} pairs/GPU STA two-view forward (BASELINE configs[1])",
                          "pairs_per_gpu": B, "H": H, "W": W_, "precision": args.precision,
                          "parallelism": (f"pair-sharded x{world}, one RCCL all-gather of the compact outputs per step on its own stream, "
                                          f"overlapping the next step's forward") if use_dist else "single GPU"},
               "gflop_per_pair": round(flops_pair / 1e9, 2),
               "whole_path_tflops": round(pairs * flops_pair / dt / 1e12, 1),
               "workspace_gb": round(model.workspace_bytes() / 1e9, 2),
               # sta_alloc_stats: (allocations / frees / stream + event creations, device-wide synchronisations) of the library's compute
               # entry points - everything was reserved before the first step (sta_reserve), the timed region adds nothing
               "hidden_allocations": {"after_sta_reserve": list(alloc_reserved), "added_by_warmup_and_survey": [alloc_before[0] - alloc_reserved[0], alloc_before[1] - alloc_reserved[1]],
                                      "added_inside_timed_region": [alloc_after[0] - alloc_before[0], alloc_after[1] - alloc_before[1]]},
               "roofline": roof}
        if use_dist:
            res.update(dist_fields)
        if world == 1 and not args.no_slam_probe:
            res["slam_224_b1"] = slam_probe(model, dev)
            if args.slam_frames > 0:
                res["slam_replay"] = slam_replay(model, dev, frames=args.slam_frames)
        if world == 1 and not args.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline_subprocess(224, 224) if args.cpu_baseline_224 else cpu_baseline_subprocess(H, W_)
        sys.stdout.flush()
        os.write(result_fd, (json.dumps(res) + "\n").encode())
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
