"""The keyframe-sequence protocol shared by the golden generator (oracle/gen_golden.py `gen_seq`) and the tests that replay
the `seq_*.npz` fixtures: which frames, which edges.  TEST INFRASTRUCTURE (no reference import, no product import)."""
import numpy as np


def seq_edge_list(i, neighbor_edge_num, loop_edge_num, loop_dist_min):
    """Edges of keyframe i in OnlineSLAM.step's order (slam.py:262-277): the <= neighbor_edge_num previous views, then the
    <= loop_edge_num loop candidates.  The reference's candidates come from `LoopDetector.detect_loop(img_gray,
    farthest_neighbor)` (DBoW3 on ORB features: a CPU stage, out of scope, absent here); the replay substitutes a deterministic
    list with the detector's own filter shape - views older than the farthest neighbour and more than `loop_dist_min` keyframes
    back (configs: 40; the short replays use 3) - ordered by a hash of (i, j) in place of the BoW similarity.
    -> (list of j, farthest neighbour)"""
    far = max(0, i - neighbor_edge_num)
    js = list(range(far, i))
    cand = [j for j in range(far) if i - j > loop_dist_min]
    cand.sort(key=lambda j: ((i * 7919 + j * 104729 + 13) % 1009, j))
    return js + cand[:loop_edge_num], far


def seq_frames(W, nkf, H, W_, seed, tag):
    """[nkf,3,H,W] normalised frames: white-noise keyframes (even) and smooth ones (odd) - two input statistics in one sequence.
    W = vista_slam_amd.weights (procedural generators)."""
    noise = W.synth_images(nkf, H, W_, seed=seed, tag=tag)
    smooth = W.smooth_images(nkf, H, W_, seed=seed, tag=tag)
    return np.stack([noise[k] if k % 2 == 0 else smooth[k] for k in range(nkf)]).astype(np.float32)
