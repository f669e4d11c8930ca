"""bench.py's contract when no measurement is possible (CPU box, too few devices): ONE JSON line with an `error` field and
a non-zero exit code -- never a bare exit (`python bench.py --gpus 8` used to return rc 1 and print
nothing).  The self-launching N > 1 path itself needs a GPU: tests/test_gpu_ddp.py::test_bench_self_launch_clean_env."""
import json
import os
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(torch.cuda.is_available(), reason="checks the no-device behaviour")
@pytest.mark.parametrize("gpus", [1, 8])
def test_bench_without_devices_prints_an_error_line(gpus):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(gpus), "--steps", "20", "--warmup", "5"], env=env,
                       capture_output=True, timeout=300)
    assert r.returncode != 0
    lines = [l for l in r.stdout.decode().splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout.decode()
    d = json.loads(lines[0])
    assert d["value"] is None and d["n_gpus"] == gpus and "error" in d and d["devices_visible"] == 0
    assert d["metric"].startswith("clips/s") and d["steps"] == 20 and d["warmup"] == 5


def test_bench_world_size_mismatch_is_reported():
    """started by a launcher whose WORLD_SIZE disagrees with --gpus: an error line from rank 0, not a hang or a bare exit"""
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29999")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--steps", "2", "--warmup", "1"], env=env,
                       capture_output=True, timeout=300)
    assert r.returncode != 0
    lines = [l for l in r.stdout.decode().splitlines() if l.startswith("{")]
    assert len(lines) == 1 and "WORLD_SIZE" in json.loads(lines[0])["error"]


@pytest.mark.skipif(torch.cuda.is_available(), reason="checks the no-device behaviour")
def test_bench_sweep_prints_one_line_per_gpu_count():
    """`--sweep-gpus 1,2,4,8`: one child per N, one JSON line per N in order (here: error lines, no device) and a non-zero exit."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--sweep-gpus", "1,2,4,8", "--steps", "3", "--warmup", "1"], env=env,
                       capture_output=True, timeout=600)
    assert r.returncode != 0
    lines = [json.loads(l) for l in r.stdout.decode().splitlines() if l.startswith("{")]
    assert [d["n_gpus"] for d in lines] == [1, 2, 4, 8] and all(d["value"] is None and "error" in d and d["steps"] == 3 for d in lines)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--sweep-gpus", "1,x"], env=env, capture_output=True, timeout=300)
    assert r.returncode == 2 and "comma-separated" in json.loads(r.stdout.decode().strip().splitlines()[-1])["error"]


def test_instrumented_steps_stay_away_from_the_barrier():
    """bench.profiled_steps: one timed step in ten carries per-launch events, never step 0 of a run with more than one step (it
    starts on an empty queue and its first brackets would carry the host's launch latency)"""
    import bench
    assert list(bench.profiled_steps(20)) == [5, 15]
    assert list(bench.profiled_steps(5)) == [4] and list(bench.profiled_steps(2)) == [1] and list(bench.profiled_steps(1)) == [0]
    assert len(bench.profiled_steps(400)) == 40 and 0 not in bench.profiled_steps(400)
