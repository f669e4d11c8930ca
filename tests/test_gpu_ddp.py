"""Data parallelism executed for real: two ranks of the training step (HIP kernels, per-block gradient buckets launched
from passt_backward's callbacks, per-bucket AdamW) against ONE process on the concatenated batch.  Both ranks share the
test box's single GPU, so the transport is gloo on device tensors (RCCL needs one device per rank); everything above
torch.distributed is the code the 8-GPU run uses.  Reference behaviour: Lightning DDP, ex_audioset.py:475-524 (mean of the
per-rank gradients, identical replicas from rank 0's initial weights)."""
import os
import socket
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORKER = os.path.join(ROOT, "tests", "ddp_worker.py")


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run(out, world, extra=()):
    port = _free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, WORKER, "--out", out, *extra], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT))
    logs = []
    for p in procs:
        try:
            o, _ = p.communicate(timeout=420)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        logs.append(o.decode(errors="replace")[-2000:])
    assert all(p.returncode == 0 for p in procs), "\n".join(logs)
    return torch.load(out)


def _rel(a, b):
    return float((a.double() - b.double()).abs().max() / b.double().abs().max())


@pytest.mark.parametrize("variant", ["fp32", "fp32+overlap_wgrad", "bf16_wire", "fp32+adamw"])
def test_two_ranks_equal_one_process_on_the_concatenated_batch(tmp_path, variant):
    opt = ("--optimizer", "adamw") if variant == "fp32+adamw" else ()
    ref = _run(str(tmp_path / "ref.pt"), 1, opt)
    extra = {"fp32": (), "fp32+overlap_wgrad": ("--overlap-wgrad",), "bf16_wire": ("--comm-dtype", "bf16"),
             "fp32+adamw": ()}[variant]
    dp = _run(str(tmp_path / "dp.pt"), 2, extra + opt)
    assert dp["world"] == 2
    # replicas identical after the steps (rank 1 started from perturbed weights: the constructor's broadcast fixed it)
    assert all(same for same, _ in dp["flags"])
    assert torch.equal(dp["init"], ref["init"])
    # mean of the two half-batch losses = the full-batch loss, step by step
    for step, full in enumerate(ref["flags"][0][1]):
        halves = [l[step] for _, l in dp["flags"]]
        assert abs(sum(halves) / 2 - full) < (5e-6 if variant != "bf16_wire" else 5e-4), (step, halves, full)
    # compare the parameter UPDATES (2 steps).  SGD is linear in the gradient: f32 wire = f32 round-off of a different
    # summation order; bf16 wire = 8-bit sums.  AdamW divides by sqrt(v): entries whose gradient is pure round-off can
    # flip sign, so it is judged in the L2 norm.
    d_dp, d_ref = (dp["params"] - dp["init"]).double(), (ref["params"] - ref["init"]).double()
    assert float(d_ref.abs().max()) > 0
    if variant == "fp32+adamw":
        e = float((d_dp - d_ref).norm() / d_ref.norm())
        assert e < 2e-2, e
    else:
        e = float((d_dp - d_ref).abs().max() / d_ref.abs().max())
        assert e < (1e-4 if variant != "bf16_wire" else 1e-2), e


@pytest.mark.parametrize("path,wire", [("attach", "fp32"), ("attach", "bf16"), ("torch_ddp", "fp32")])
def test_autograd_path_two_ranks_equal_one_process(tmp_path, path, wire):
    """The DROP-IN path -- net(x); loss.backward(); torch optimizer, what an unmodified ex_audioset.py runs (:179-186) --
    under data parallelism, two ways: passt_amd.ddp.attach(net) (the autograd node all-reduces per-block buckets from
    inside its backward and returns averaged gradients) and torch's DistributedDataParallel(find_unused_parameters=True)
    around the same module (ex_audioset.py:488-489 via Lightning).  Both against ONE process of the *TrainStep* path on the
    concatenated batch: ties the two product paths and the two DDP mechanisms together."""
    ref = _run(str(tmp_path / "ref.pt"), 1)
    ref_auto = _run(str(tmp_path / "ref_auto.pt"), 1, ("--path", "attach"))
    # one process: autograd path == TrainStep path (same kernels, torch SGD vs pa_sgd)
    d_a, d_r = (ref_auto["params"] - ref_auto["init"]).double(), (ref["params"] - ref["init"]).double()
    assert float((d_a - d_r).abs().max() / d_r.abs().max()) < 1e-5
    extra = ("--path", path) + (("--comm-dtype", "bf16") if wire == "bf16" else ())
    dp = _run(str(tmp_path / "dp.pt"), 2, extra)
    assert dp["world"] == 2
    _check_against_single_process(dp, ref, wire)


def test_bench_self_launch_clean_env():
    """`python bench.py --gpus 2 ...` exactly as the driver types it for N = 1 -- NO launcher, NO RANK / WORLD_SIZE in the
    environment: bench.py forks its own ranks (ex_audioset.py:499-524 does the same for DDP=N) and rank 0's line comes out
    of the parent.  On the one-GPU test box the ranks share device 0 over gloo (PASST_AMD_BENCH_DRY_GLOO=1, labelled in the
    line); on a multi-GPU box the same command runs over RCCL.  Both product paths."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["PASST_AMD_BENCH_DRY_GLOO"] = "1"
    for path in ("trainstep", "autograd"):
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "4",
                            "--path", path], env=env, capture_output=True, timeout=900)
        out = r.stdout.decode()
        assert r.returncode == 0, (out, r.stderr.decode()[-3000:])
        lines = [l for l in out.splitlines() if l.startswith("{")]
        assert len(lines) == 1, out
        d = json.loads(lines[0])
        assert "error" not in d and d["n_gpus"] == 2 and d["config"]["global_batch"] == 8 and d["value"] > 0
        assert "DRY RUN" in d["config"]["parallelism"]
        ar = d["allreduce_measured"]
        assert len(ar["buckets"]) == 14 and len(ar["idle"]["buckets"]) == 14 and ar["idle"]["ms_per_step"] > 0
        assert d["rccl"]["nranks"] == 2 and [x["rank"] for x in d["rccl"]["ranks"]] == [0, 1]


def _check_against_single_process(dp, ref, variant):
    assert all(same for same, _ in dp["flags"])
    assert torch.equal(dp["init"], ref["init"])
    world = dp["world"]
    for step, full in enumerate(ref["flags"][0][1]):
        parts = [l[step] for _, l in dp["flags"]]
        assert abs(sum(parts) / world - full) < (5e-6 if "bf16" not in variant else 5e-4), (step, parts, full)
    d_dp, d_ref = (dp["params"] - dp["init"]).double(), (ref["params"] - ref["init"]).double()
    e = float((d_dp - d_ref).abs().max() / d_ref.abs().max())
    assert e < (1e-4 if "bf16" not in variant else 1e-2), e


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="RCCL needs one device per rank: runs on the first multi-GPU box")
@pytest.mark.parametrize("path", ["trainstep", "attach"])
@pytest.mark.parametrize("transport", ["torch", "rccl_abi"])
@pytest.mark.parametrize("wire", ["fp32", "bf16"])
def test_ranks_over_rccl_equal_one_process(tmp_path, transport, wire, path):
    """The same comparison over the REAL wire: backend nccl (= RCCL over xGMI), one device per rank, through
    torch.distributed and through the library's own pa_comm_* entry points; 2 ranks and -- when the box has them -- 4 / 8
    (global batch 8); TrainStep and the drop-in path with ddp.attach.  Skipped on a single-GPU box, self-verifying on the first
    node that has more."""
    ref = _run(str(tmp_path / "ref.pt"), 1)
    worlds = [w for w in (2, 4, 8) if w <= torch.cuda.device_count()]
    for world in worlds:
        extra = ("--backend", "nccl", "--transport", transport, "--path", path) + (("--comm-dtype", "bf16") if wire == "bf16" else ())
        dp = _run(str(tmp_path / f"dp{world}.pt"), world, extra)
        assert dp["world"] == world
        # the communicator itself spans `world` devices: ncclCommCount through pa_comm_info (C-ABI transport) / the process group
        assert dp["comm_info"] is not None and dp["comm_info"]["nranks"] == world and dp["comm_info"]["rank"] == 0, dp["comm_info"]
        if transport == "rccl_abi":
            assert "rccl_version" in dp["comm_info"] and "C ABI" in dp["comm_info"]["backend"]
        _check_against_single_process(dp, ref, wire)


def test_rccl_abi_transport_single_rank():
    """The C-ABI collective entry (pa_comm_init / pa_allreduce_bucket, RCCL loaded by the library) on the one GPU a test
    box has: a world of 1 is the identity, for both wire types, ordered on the transport's own stream; a second bucket
    reuses the communicator.  (Two ranks need two devices: the 8-GPU scaling run is the driver's.)"""
    from passt_amd.ddp import RcclAbiTransport
    tr = RcclAbiTransport("cuda:0")
    assert tr.world == 1 and tr.rank == 0
    a = torch.randn(1 << 20, device="cuda")
    b = (torch.randn(4099, device="cuda")).to(torch.bfloat16)
    a0, b0 = a.clone(), b.clone()
    h1, h2 = tr.all_reduce(a), tr.all_reduce(b)
    h1.wait()
    h2.wait()
    torch.cuda.synchronize()
    assert torch.equal(a, a0) and torch.equal(b, b0)
    tr.close()


def test_bench_multi_rank_dry_run():
    """bench.py's N > 1 path end to end -- rank / world from the environment, barrier + max-over-ranks timing, the
    reducer's bucket callbacks, the MEASURED all-reduce object (HIP events per bucket), one JSON line from rank 0 -- on the
    one GPU a test box has: two ranks on device 0 over gloo (PASST_AMD_BENCH_DRY_GLOO=1).  The 8-GPU run is the driver's;
    this is what keeps it from tripping over an untested code path."""
    import json
    port = _free_port()
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY="0", PASST_AMD_BENCH_DRY_GLOO="1")
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                                       "--batch", "4"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE))
    outs = []
    for p in procs:
        o, e = p.communicate(timeout=600)
        outs.append((p.returncode, o.decode(), e.decode()[-1500:]))
    assert all(rc == 0 for rc, _, _ in outs), outs
    lines = [l for l in outs[0][1].splitlines() if l.startswith("{")]
    assert len(lines) == 1 and not [l for l in outs[1][1].splitlines() if l.startswith("{")]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["global_batch"] == 8 and "DRY RUN" in d["config"]["parallelism"]
    ar = d["allreduce_measured"]
    assert len(ar["buckets"]) == 14 and ar["bytes_per_step"] > 300e6          # head + 12 blocks + patch embedding, f32
    assert all(b["in_flight_ms"] > 0 and b["exposed_wait_ms"] >= 0 for b in ar["buckets"])


@pytest.mark.parametrize("fail_rank", [0, 1])
def test_bench_multi_rank_line_survives_failing_diagnostics(fail_rank):
    """The N > 1 diagnostics (instrumented step, idle all-reduces, device gather) run AFTER the measurement is complete: a rank
    whose diagnostics fail (injected) must not cost the line.  The failing rank leaves without another collective; the peers'
    watchdog (here 15 s) releases them; rank 0 prints value / ms_per_step with `multi_gpu_diagnostics_error`; every rank exits 0."""
    import json
    port = _free_port()
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY="0", PASST_AMD_BENCH_DRY_GLOO="1", PASST_AMD_BENCH_DIAG_FAIL_RANK=str(fail_rank),
                   PASST_AMD_BENCH_DIAG_TIMEOUT_S="15")
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                                       "--batch", "4"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE))
    outs = []
    for p in procs:
        o, e = p.communicate(timeout=600)
        outs.append((p.returncode, o.decode(), e.decode()[-1500:]))
    assert all(rc == 0 for rc, _, _ in outs), outs
    lines = [l for l in outs[0][1].splitlines() if l.startswith("{")]
    assert len(lines) == 1, outs
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["value"] > 0 and d["ms_per_step"] > 0
    assert "multi_gpu_diagnostics_error" in d and "allreduce_measured" not in d


def test_bench_sweep_dry_run():
    """`python bench.py --sweep-gpus 1,2` -- the single command that yields the weak-scaling curve on a multi-GPU node -- on the one
    GPU of a test box: N = 1 for real, N = 2 as two self-launched ranks on device 0 over gloo (PASST_AMD_BENCH_DRY_GLOO=1).  One
    JSON line per N, in order; the N = 2 line leads with the idle all-reduce figure and carries the communicator evidence; the
    scaling model says which bus efficiency it used (the dry run must NOT feed its gloo figure into it)."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(HSA_ENABLE_IPC_MODE_LEGACY="0", PASST_AMD_BENCH_DRY_GLOO="1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--sweep-gpus", "1,2", "--steps", "2", "--warmup", "1", "--batch", "4",
                        "--no-cpu-baseline"], env=env, capture_output=True, timeout=900)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    lines = [json.loads(l) for l in r.stdout.decode().splitlines() if l.startswith("{")]
    assert [d["n_gpus"] for d in lines] == [1, 2] and all(d["value"] > 0 for d in lines)
    one, two = lines
    assert one["config"]["global_batch"] == 4 and two["config"]["global_batch"] == 8 and "roofline" in one
    assert two["rccl_nranks"] == 2 and two["allreduce_bus_GBps_idle"] == two["allreduce_measured"]["idle"]["bus_GBps_total"] > 0
    assert list(two)[:16].index("allreduce_bus_GBps_idle") < list(two).index("config")
    assert two["scaling_model"]["bus_efficiency"] == 0.35 and "ASSUMED" in two["scaling_model"]["inputs"]


def test_bench_speedtest_cli():
    """`python bench.py --speedtest` (the reference's model_speed_test flow): one JSON line per batch size with the
    dynamo / GradScaler evidence in it."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--speedtest", "--batch", "4", "--steps", "3", "--warmup", "2"],
                       env=env, capture_output=True, timeout=900)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    lines = [json.loads(l) for l in r.stdout.decode().splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = lines[0]
    assert d["unit"] == "specs/s" and d["value"] > 0 and d["config"]["per_gpu_batch"] == 4 and d["config"]["compiled"] is True
    assert d["grad_scale_after"] == 65536.0 and d["dynamo"]["graphs_captured"] == 0 and d["logits_dtype"] == "torch.float32"
    assert d["dynamo"]["frames_total"] == d["dynamo"]["frames_after_warmup"] <= 1
