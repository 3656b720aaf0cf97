"""SLAM-scale stage under rocprofv3 (--kernel-trace --stats): python tools/slam_trace.py encode|sched|replay [reps]
prints the wall time per call so that the sum of kernel durations can be compared with it (idle gaps).  replay: the three-stream
pipelined bench.slam_replay (reps = frames)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from vista_slam_amd import weights as W
from vista_slam_amd.sta_frontend import STAFrontend
from vista_slam_amd.slam_scheduler import regress_views
what = sys.argv[1] if len(sys.argv) > 1 else "encode"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
m = STAFrontend(W.FULL, "cuda:0").load_procedural(seed=43)
if what == "replay":
    import bench
    r = bench.slam_replay(m, "cuda:0", frames=reps, warm=8)
    print(f"replay: {r['keyframes_per_s']} keyframes/s pipelined ({r['ms_per_keyframe']} ms per keyframe), single stream {r['single_stream']['keyframes_per_s']} per call")
    sys.exit(0)
imgs = torch.from_numpy(W.synth_images(2, 224, 224, seed=43, tag=7)).cuda()
fa, pa = m._encode_image(imgs[:1], None, normalize=False)
fb, pb = m._encode_image(imgs[1:], None, normalize=False)
fn = (lambda: m._encode_image(imgs[:1], None, normalize=False)) if what == "encode" else \
     (lambda: regress_views(m, fa, [fb] * 5, [True] * 5, 0.0, 224, 224))
for _ in range(3):
    fn()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(reps):
    fn()
torch.cuda.synchronize()
print(f"{what}: {(time.perf_counter() - t0) / reps * 1e3:.3f} ms per call over {reps} calls (+3 warm-up, +2 setup encodes)")
