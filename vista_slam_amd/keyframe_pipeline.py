"""The frontend calls of `OnlineSLAM.step` (vista_slam/slam.py:244-297) over a keyframe sequence, in three schedules.

Per keyframe i the SLAM loop does `add_view(i)` (slam.py:142-151: encode, cache the features) and then `connect_view_i_j(i, j)`
(slam.py:191-241) for the <= neighbor_edge_num previous views and the <= loop_edge_num loop candidates: `regress_two_views`
(slam.py:153-189) and, for an accepted edge, the node bookkeeping (slam.py:203-218: a node per view; for a view that already
has a node, the scale edge to its FIRST node = `estimate_scale_with_depth_and_confidence` + the sqrt-mean confidence).
`replay()` issues exactly those calls against a growing feature cache and returns one `EdgeRecord` per candidate edge, in the
reference's edge order - the same records whichever schedule produced them:

  "split"      the four split calls per edge, B = 1, as slam.py:162-185 issues them (`_decode_stereo`, `head_pose_s`, a host
               read of the confidence, `head_pts` x 2, `estimate_intrinsic_from_pts3d(shared_intrinsic=True)`, depths): the
               zero-edit drop-in path;
  "batched"    all candidate edges of a keyframe in one native scheduler call (`slam_scheduler.regress_views`, row f2);
  "pipelined"  three streams - add_view(i+1) | decode + pose heads of keyframe i's edges (`regress_views_begin`) | DPT heads,
               reductions and bookkeeping of keyframe i-1 (`regress_views_finish`) - using only independence the loop itself
               has: add_view(i+1) needs no result of keyframe i (slam.py:258) and the edges of keyframe i need encoder
               features only (slam.py:153-162).

What is NOT here is the reference's CPU side of the loop (pypose Sim3 bookkeeping, DBoW3 loop detection, PGO): the caller
supplies `edge_list(i)`.  bench.py's `slam_replay` and tests/test_gpu_parity.py (the `seq_*` reference goldens) both run this.
All arithmetic is in libsta_mi355.so.
"""
from __future__ import annotations

from typing import Callable, Dict, List, Optional, Sequence, Tuple

import torch

from .post import estimate_intrinsic_from_pts3d, estimate_scale_with_depth_and_confidence
from .slam_scheduler import EdgeResult, regress_views, regress_views_begin, regress_views_finish
from .sta_frontend import STAFrontend

SCHEDULES = ("split", "batched", "pipelined")


class EdgeRecord:
    """One candidate edge (i, j): what `regress_two_views` returned + the scale edges `connect_view_i_j` derived from it.
    scales / scale_confs: [for view i, for view j]; None where that view got its first node from this edge (or the edge was
    rejected)."""
    __slots__ = ("i", "j", "pose", "rel_pose_conf", "accepted", "confs", "intri", "depths", "scales", "scale_confs")

    def __init__(self, i, j, r: EdgeResult):
        self.i, self.j = i, j
        self.pose, self.rel_pose_conf, self.accepted = r.pose, r.rel_pose_conf, r.accepted
        self.confs, self.intri, self.depths = r.confs, r.intri, r.depths
        self.scales: List[Optional[torch.Tensor]] = [None, None]
        self.scale_confs: List[Optional[torch.Tensor]] = [None, None]


class NodeBook:
    """Device arithmetic of the node bookkeeping in `connect_view_i_j` (slam.py:203-218)."""

    def __init__(self, frontend: STAFrontend):
        self.frontend = frontend
        self.first: Dict[int, Tuple[torch.Tensor, torch.Tensor, torch.Tensor]] = {}      # view -> (depth, conf, intri) of its first node
        self.scale_edges = 0

    def add(self, rec: EdgeRecord):
        if not rec.accepted:
            return
        for k, v in enumerate((rec.i, rec.j)):
            depth, conf = rec.depths[k], rec.confs[k]
            if v in self.first:
                d0, c0, _K = self.first[v]
                rec.scales[k] = estimate_scale_with_depth_and_confidence(self.frontend, depth, d0, conf, c0)
                rec.scale_confs[k] = (conf * c0).sqrt().mean()
                self.scale_edges += 1
            else:
                # copies, not views: depth / conf are slices of the scheduler call's whole [k, 2, H, W] outputs and intri of its
                # [k, 3, 3] - kept as views a view's first node would pin all k edges' maps for the rest of the run (with
                # keep_records=False up to 6x the memory a keyframe needs).  Cloned on the current stream, i.e. behind the wait
                # for the edge stream's `after` event in the pipelined schedule.
                self.first[v] = (depth.clone(), conf.clone(), rec.intri.clone())


def regress_two_views_split(frontend: STAFrontend, feat_i, feat_j, pos_i, pos_j, adjacent: bool, rel_pose_thres: float,
                            H: int, W: int) -> EdgeResult:
    """One edge (i, j) through the four split entry points, in the order `OnlineSLAM.regress_two_views` uses them
    (slam.py:153-189): decode both directions, pose head on the i -> j token, the host-side accept / reject test (:169), then the
    point-map head on the j side and on the i side, the pair-shared intrinsics and the depths."""
    tokens_i, tokens_j = frontend._decode_stereo(feat_i, feat_j, pos_i, pos_j)
    head = frontend.head_pose_s(tokens_i[-1][:, 0, :])
    conf = float(head["conf"][0])                                     # the one host read per edge (the reference compares on the host too)
    if not adjacent and conf < rel_pose_thres:
        return EdgeResult(head["pose"][0], conf, False)
    shape = [[H, W]]
    maps = {}
    for side, feat, toks in (("j", feat_j, tokens_j), ("i", feat_i, tokens_i)):      # j side first, like slam.py:179-180
        maps[side] = frontend.head_pts([feat] + [t[:, 1:, :] for t in toks], shape)
    pts = torch.cat([maps["i"]["pts3d"], maps["j"]["pts3d"]], dim=0)               # view order [i -> j, j -> i] (slam.py:182)
    conf_maps = torch.cat([maps["i"]["conf"], maps["j"]["conf"]], dim=0)
    K = estimate_intrinsic_from_pts3d(frontend, pts, conf_maps, shared_intrinsic=True)
    return EdgeResult(head["pose"][0], conf, True, conf_maps, K, pts[..., 2], pts)


def replay(frontend: STAFrontend, n_keyframes: int, add_view: Callable[[int], Tuple[torch.Tensor, Optional[torch.Tensor]]],
           edge_list: Callable[[int], Sequence[int]], rel_pose_thres: float, H: int, W: int, schedule: str = "batched",
           streams: Optional[Sequence[torch.cuda.Stream]] = None, timeline: Optional[Dict[str, list]] = None,
           on_edges: Optional[Callable[[int, Sequence[int], List[EdgeRecord]], None]] = None, keep_records: bool = True):
    """Run keyframes 0 .. n_keyframes-1.  `add_view(i)` enqueues the encode of keyframe i on the CURRENT stream and returns
    (feature [1,N,E], positions or None); `edge_list(i)` -> the views j < i to connect, in the reference's order.
    Returns (records, book, feats): every candidate edge as an EdgeRecord, the NodeBook, the feature cache.
    `streams` (pipelined): [encode stream, edge stream 0, edge stream 1]; default: `frontend.pipeline_streams(3)`, three
    library-owned streams measured to overlap pairwise (which application streams share a hardware queue is not visible through
    the HIP API).
    `timeline`: optional dict of lists that receives (start, end) CUDA-event pairs per stage ("edges_decode", "edges_heads",
    "bookkeeping"; "edges" for the un-split schedules).  `on_edges(i, js, records)`: called after keyframe i's bookkeeping was
    enqueued (e.g. pose chaining in the harness).  keep_records=False: the per-edge maps are dropped once `on_edges` has seen
    them (a long replay would otherwise hold every edge's point maps)."""
    assert schedule in SCHEDULES, schedule
    dev = frontend.device
    feats: List[torch.Tensor] = []
    poss: List[Optional[torch.Tensor]] = []
    book = NodeBook(frontend)
    records: List[EdgeRecord] = []

    def mark():
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        return e

    def span(key, t0):
        if timeline is not None:
            timeline.setdefault(key, []).append((t0, mark()))

    def consume(i, js, results):
        t0 = mark() if timeline is not None else None
        recs = [EdgeRecord(i, j, r) for j, r in zip(js, results)]
        for rec in recs:
            book.add(rec)
        if keep_records:
            records.extend(recs)
        if on_edges is not None:
            on_edges(i, js, recs)
        if timeline is not None:
            span("bookkeeping", t0)

    if schedule != "pipelined":
        for i in range(n_keyframes):
            feat, pos = add_view(i)
            feats.append(feat); poss.append(pos)
            js = list(edge_list(i))
            if not js:
                continue
            t0 = mark() if timeline is not None else None
            if schedule == "split":
                def pos_of(v):
                    return poss[v] if poss[v] is not None else frontend._positions(1, H // 16, W // 16)
                res = [regress_two_views_split(frontend, feats[i], feats[j], pos_of(i), pos_of(j), i - j == 1, rel_pose_thres, H, W) for j in js]
            elif timeline is None:
                res = regress_views(frontend, feat, [feats[j] for j in js], [i - j == 1 for j in js], rel_pose_thres, H, W)
            else:                                        # the same call as its two phases, with a mark between them
                pend = regress_views_begin(frontend, feat, [feats[j] for j in js], H, W)
                span("edges_decode", t0)
                t0 = mark()
                res = regress_views_finish(frontend, pend, [i - j == 1 for j in js], rel_pose_thres)
                span("edges_heads", t0)
            if timeline is not None and schedule == "split":
                span("edges", t0)
            consume(i, js, res)
        return records, book, feats

    # ---- pipelined: three streams
    main_stream = torch.cuda.current_stream(dev)
    if streams is None:
        streams = frontend.pipeline_streams(3)
    enc_stream, edge_streams = streams[0], [streams[1], streams[2]]
    for s in streams:
        s.wait_stream(main_stream)                       # whatever the caller enqueued before (frames, weights) is visible to the lanes

    def begin(i, feat):                                  # on the CURRENT stream
        js = list(edge_list(i))
        if not js:
            return None
        t0 = mark() if timeline is not None else None
        pend = regress_views_begin(frontend, feat, [feats[j] for j in js], H, W)
        if timeline is not None:
            span("edges_decode", t0)
        return i, js, pend

    def finish(job, after):                              # on the stream begin(i) ran on
        i, js, pend = job
        t0 = mark() if timeline is not None else None
        res = regress_views_finish(frontend, pend, [i - j == 1 for j in js], rel_pose_thres)
        if timeline is not None:
            span("edges_heads", t0)
        if after is not None:                            # the previous keyframe's bookkeeping ran on the other edge stream
            torch.cuda.current_stream(dev).wait_event(after)
        consume(i, js, res)
        done = torch.cuda.Event()
        done.record()
        return done

    def add_view_async(i):
        with torch.cuda.stream(enc_stream):
            feat, pos = add_view(i)
            done = torch.cuda.Event()
            done.record()
        return feat, pos, done

    nxt = add_view_async(0)
    job, last_fin = None, None
    opened = []                                          # every scheduler call begun (closed again in `finally`: close() is idempotent)
    try:
        for i in range(n_keyframes):
            feat, pos, done = nxt
            feats.append(feat); poss.append(pos)
            if i + 1 < n_keyframes:
                nxt = add_view_async(i + 1)
            with torch.cuda.stream(edge_streams[i & 1]):
                edge_streams[i & 1].wait_event(done)
                new_job = begin(i, feat)
                if new_job is not None:
                    opened.append(new_job[2])
            if job is not None:
                with torch.cuda.stream(edge_streams[(i - 1) & 1]):
                    last_fin = finish(job, last_fin)
            job = new_job
            del opened[:-2]
        if job is not None:
            with torch.cuda.stream(edge_streams[(n_keyframes - 1) & 1]):
                last_fin = finish(job, last_fin)
    finally:
        for pend in opened:                              # an exception between the phases must not leave a stream's context reserved
            pend.close()
    if last_fin is not None:
        main_stream.wait_event(last_fin)
    main_stream.wait_stream(enc_stream)
    for s in edge_streams:
        main_stream.wait_stream(s)
    return records, book, feats
