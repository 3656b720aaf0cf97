"""CPU: bench.py's launcher contract - `python bench.py --gpus N` from a bare shell must start its own N ranks
(the round-1 script died on an assert unless it was launched under torch.distributed.run)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_self_launches_ranks_when_world_size_is_unset(tmp_path):
    """A stand-in `torch.distributed.run` (first on PYTHONPATH) records the command line bench.py re-executes itself with."""
    pkg = tmp_path / "torch" / "distributed"
    pkg.mkdir(parents=True)
    (tmp_path / "torch" / "__init__.py").write_text("")
    (pkg / "__init__.py").write_text("")
    (pkg / "run.py").write_text("import sys, json\nprint('LAUNCH ' + json.dumps(sys.argv[1:]))\n")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["PYTHONPATH"] = str(tmp_path)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1"],
                       env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    line = [l for l in r.stdout.splitlines() if l.startswith("LAUNCH ")][0]
    args = line[len("LAUNCH "):]
    assert "--nproc-per-node=2" in args and "127.0.0.1" in args and "bench.py" in args
    assert '"--gpus", "2"' in args and '"--steps", "3"' in args


def test_bench_under_a_launcher_does_not_relaunch():
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert 'args.gpus > 1 and "WORLD_SIZE" not in os.environ' in src
