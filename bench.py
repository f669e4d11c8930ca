#!/usr/bin/env python
"""bench.py -- headline benchmark of the PaSST training hot path on MI355X.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python bench.py --gpus N --steps K --warmup W            (self-launching: forks one rank per GPU, like the reference's
                                                              DDP=N python ex_audioset.py, ex_audioset.py:499-524)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W   (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the env)
Whatever goes wrong (fewer than N devices, a rank dying), the process prints ONE JSON line with an "error" field and exits
non-zero -- never a bare exit.

Metric (BASELINE.json): clips/s of 10 s @ 32 kHz, forward+backward, passt_s
(passt_s_swa_p16_128_ap476: 768/12/12, patch 16 stride 10, s_patchout_t=40 s_patchout_f=4 => 474
tokens), batch 64 per GPU (configs[1]; weak scaling: per-GPU batch fixed).  One "step" = one full
training step on one resident synthetic batch: waveforms (64, 320000) -> fused mel front end (train
mode: random fmin/fmax, SpecAugment masks) -> mixup -> PaSST forward (bf16 MFMA, f32 residual stream)
-> BCE loss -> full backward -> [N>1: per-block RCCL all-reduce overlapped with backward] -> AdamW on
all 86 M parameters.  Nothing is skipped or cached inside the timed region.

Besides the contract line it reports
  roofline     : the dominant kernel (the MFMA GEMM family), timed per launch with HIP events on the
                 launch stream inside the timed region; achieved = algorithmic FLOPs / measured time,
                 against the 2.5 PFLOP/s dense bf16 MFMA peak (MI355X_MICROARCH.md).
  attention    : the fused attention kernels, timed the same way in the same run (MFMA utilisation).
  frontend     : the mel front-end kernel: algorithmic GB/s against the 8 TB/s HBM peak.
  scaling_model: a MODELLED 2/4/8-GPU curve from this run's step time and the reducer's bucket sizes
                 (the measured curve is the driver's SCALE_rNN.json when it has an 8-GPU node).
--config c4 | c4_ref | c5 runs the other BASELINE.json configurations (ViT-L geometry, the reference's passt_l,
ESC-50 fine-tune); their lines are committed under profiles/.
  cpu_baseline : the oracle (CPU restatement of the reference, oracle/passt_oracle.py) timed on this
                 host on a bounded sample of the same workload (train-mode fwd+bwd, B=2 per iteration).
"""
import argparse
import contextlib
import json
import os
import sys
import threading
import time
import warnings

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

BF16_MFMA_PEAK_TFLOPS = 2500.0      # dense, /opt/skills/guides/MI355X_MICROARCH.md
HBM_PEAK_GBPS = 8000.0              # same guide
ARCH = "passt_s_swa_p16_128_ap476"
CLIP_SAMPLES = 320000                # 10 s @ 32 kHz

# BASELINE.json configs.  The driver's line (no --config) is c2, the configuration the metric is quoted on; the others
# are run by hand and their lines are committed under profiles/ (bench lines, not parity cases).
#   tokens = kept patches + 2; dims = (embed_dim, depth, heads)
CONFIGS = {
    "c2": dict(metric="clips/s (10s@32k) fwd+bwd passt_s", arch=ARCH, net_kw=dict(s_patchout_t=40, s_patchout_f=4), custom=None,
               n_classes=527, batch=64, clip=CLIP_SAMPLES, loss="bce", dims=(768, 12, 12), tokens=474, kept=472,
               mel_kw=dict(freqm=48, timem=192),
               desc="passt_s_swa_p16_128_ap476 train step (mel front end + mixup + fwd + BCE + bwd + {opt}), 768/12/12, "
                    "474 tokens (s_patchout_t=40,f=4), random-init weights"),
    "c4": dict(metric="clips/s (10s@32k) fwd+bwd ViT-L/16 (1024/24/16) u_patchout=400", arch=None,
               custom=dict(embed_dim=1024, depth=24, num_heads=16, u_patchout=400), net_kw={}, n_classes=527, batch=32,
               clip=CLIP_SAMPLES, loss="bce", dims=(1024, 24, 16), tokens=790, kept=788, mel_kw=dict(freqm=48, timem=192),
               desc="BASELINE config #4: ViT-L/16 geometry 1024/24/16, u_patchout=400 (790 tokens), train step (mel + mixup + "
                    "fwd + BCE + bwd + {opt}), random-init weights"),
    "c4_ref": dict(metric="clips/s (10s@32k) fwd+bwd passt_l (768/7/12) u_patchout=400", arch="passt_l_kd_p16_128_ap47",
                   net_kw=dict(u_patchout=400), custom=None, n_classes=527, batch=32, clip=CLIP_SAMPLES, loss="bce",
                   dims=(768, 7, 12), tokens=790, kept=788, mel_kw=dict(freqm=48, timem=192),
                   desc="the reference's own passt_l (passt_l_kd_p16_128_ap47: 768/7/12), u_patchout=400 (790 tokens), train "
                        "step (mel + mixup + fwd + BCE + bwd + {opt}), random-init weights"),
    "c5": dict(metric="clips/s (5s@32k) fwd+bwd passt_s ESC-50 fine-tune", arch=ARCH, net_kw=dict(s_patchout_t=10, s_patchout_f=3),
               custom=None, n_classes=50, batch=12, clip=160000, loss="ce", dims=(768, 12, 12), tokens=353, kept=351,
               mel_kw=dict(freqm=48, timem=80),
               desc="BASELINE config #5: ESC-50 fine-tune (ex_esc50.py:40,60-64): passt_s n_classes=50, 5 s clips (500 frames, "
                    "random time-pos-embed offset), s_patchout_t=10,f=3 (353 tokens), CE-mixup loss, train step incl. {opt}"),
}


DIAG_TIMEOUT_S = float(os.environ.get("PASST_AMD_BENCH_DIAG_TIMEOUT_S", "240"))   # N > 1: watchdog of the post-measurement diagnostics
PROFILE_EVERY = 10     # timed steps between two steps that carry per-launch HIP events (two event records per launch cost the
                       # instrumented step ~6 %: r04 sampled 1 step in 5, r05 samples 1 in 10 -- steps 5 and 15 of the default 20)


def profiled_steps(steps):
    """The instrumented timed steps.  Not step 0: it starts on an empty queue right behind the barrier, so the brackets of its
    first launches (the front end is THE first) carry the host's launch latency -- 113 us against the tracer's 89 for that kernel."""
    return range(min(PROFILE_EVERY // 2, steps - 1), steps, PROFILE_EVERY)


def algorithmic_gflop_per_clip(N=474, D=768, depth=12, kept_patches=472):
    """per block 24 N D^2 + 4 N^2 D, patch embed 2 P 256 D (kept patches only),
    fwd+bwd = 3x fwd."""
    fwd = depth * (24 * N * D * D + 4 * N * N * D) + 2 * kept_patches * 256 * D
    return 3 * fwd / 1e9


def skipped_tail_gflop_per_clip(N=474, D=768):
    """FLOPs of the reference algorithm the prefix-only tail does NOT execute: the last block's proj / fc1 / fc2
    on the N - 2 non-prefix rows (18 D^2 per row forward) and its attention for the N - 2 non-prefix queries (4 N D per query),
    forward + backward = 3 x forward."""
    return 3 * (18.0 * (N - 2) * D * D + 4.0 * (N - 2) * N * D) / 1e9


def modelled_scaling(ms_step_1gpu, bucket_bytes, link_gbps=153.0, links=7, bus_eff=0.35):
    """MODELLED (not measured) weak-scaling curve for N = 2, 4, 8 GPUs of one node from the measured single-GPU step and
    the measured per-bucket wire bytes: ring all-reduce of S bytes moves 2 (N-1)/N S per GPU; xGMI is point-to-point,
    a ring uses one link per direction, RCCL runs one ring per link it can (min(N-1, links) of 153 GB/s each).  Every
    bucket but the LAST overlaps with the backward; the last one is exposed, minus the optimizer launches of the earlier
    buckets that now run under it (TrainStep drains buckets in order).  Efficiency = t1 / tN."""
    sizes = list(bucket_bytes.values())
    total, last = sum(sizes), sizes[-1] + (sizes[-2] if len(sizes) > 1 else 0)     # block 0 and patch-embedding buckets
    out = {}
    for n in (2, 4, 8):
        # bus_eff: RCCL's large-message bus bandwidth as a fraction of the aggregate link peak -- 0.35 is an ASSUMPTION; an N > 1
        # run replaces it with what its own idle all-reduce of these buckets measured (allreduce_measured.idle.bus_GBps_total)
        bw = min(n - 1, links) * link_gbps * 1e9 * bus_eff
        t_all = 2.0 * (n - 1) / n * total / bw * 1e3            # ms on the wire per step
        t_last = 2.0 * (n - 1) / n * last / bw * 1e3
        bwd_window = 0.6 * ms_step_1gpu                         # the backward is ~60 % of the step
        exposed = max(0.0, t_all - t_last - bwd_window) + max(0.0, t_last - 0.4)   # 0.4 ms of optimizer work covers the tail
        # while a bucket is on the wire the backward's MFMA kernels (~85 % of its time) share the CUs with the all-reduce kernels:
        # measured on one GPU with a stand-in that holds whole CUs (profiles/r04_copersist_probe.txt): +28 % per GEMM from 8 CUs on
        # (nothing if RCCL's workgroups fit beside ours); plus the 1.6 % the one-item-per-workgroup launch form costs the GEMMs
        interference = 0.28 * 0.85 * min(t_all, bwd_window) + 0.016 * 0.7 * ms_step_1gpu
        out[str(n)] = {"wire_ms": round(t_all, 3), "exposed_ms": round(exposed, 3), "interference_ms": round(interference, 3),
                       "efficiency": round(ms_step_1gpu / (ms_step_1gpu + exposed + interference), 4)}
    return out


def cpu_baseline(budget_s=12.0, threads=None):
    """Oracle (port of the reference) fwd+bwd in train mode on this host's cores.  Bounded: at most
    ~budget_s of CPU work; the thread count is capped (torch's CPU GEMMs stop scaling -- and collapse from
    oversubscription -- long before the 100+ hardware threads of a GPU host)."""
    from oracle import detgen
    from oracle import passt_oracle as O
    threads = threads or min(32, os.cpu_count() or 1)
    torch.set_num_threads(threads)
    cfg = O.make_cfg(s_patchout_t=40, s_patchout_f=4)
    sd = O.to_torch(detgen.passt_state_dict(cfg, 1), requires_grad=True)
    B = 2
    x = torch.from_numpy(detgen.uniform(1, "x", (B, 1, 128, 998), -1, 1))
    y = (torch.rand(B, 527) < 0.005).float()
    times = []
    t_start = time.time()
    while True:
        t0 = time.time()
        lo, _ = O.passt_forward(sd, x, cfg, training=True)
        O.bce_loss(lo, y).backward()
        times.append(time.time() - t0)
        if time.time() - t_start > budget_s or len(times) >= 40:
            break
    used = times[1:] if len(times) > 1 else times          # drop the warm-up iteration when there is another
    dt = sum(used)
    return {"value": round(len(used) * B / dt, 3), "unit": "clips/s", "cores": threads, "kind": "port",
            "sample": f"{len(used)} train-mode fwd+bwd iterations of batch {B} (net only, 474 tokens, fp32 torch CPU "
                      f"restatement of the reference, {threads} threads of {os.cpu_count()} hw threads), {dt:.1f} s"}


def cpu_baseline_forward(budget_s=10.0, threads=None):
    """north_star: "the reference CPU forward timed on the same host (core count stated)" = BASELINE config #1 exactly:
    passt_s_swa_p16_128_ap476, eval-mode forward of batch 2 synthetic 10 s @ 32 kHz waveforms, mel front end + network,
    no patchout (1190 tokens), fp32, on the oracle (the port of the reference; /root/reference is not on the GPU box)."""
    from oracle import detgen
    from oracle import passt_oracle as O
    threads = threads or min(32, os.cpu_count() or 1)
    torch.set_num_threads(threads)
    cfg = O.make_cfg()
    sd = O.to_torch(detgen.passt_state_dict(cfg, 1))
    B = 2
    wave = torch.from_numpy(detgen.uniform(2, "wave", (B, CLIP_SAMPLES), -0.1, 0.1))
    times = []
    t_start = time.time()
    with torch.no_grad():
        while True:
            t0 = time.time()
            mel = O.mel_frontend(wave, training=False, fmin_aug_range=10, fmax_aug_range=2000)
            O.passt_forward(sd, mel[:, None, :, :998], cfg, training=False)
            times.append(time.time() - t0)
            if time.time() - t_start > budget_s or len(times) >= 40:
                break
    used = times[1:] if len(times) > 1 else times
    dt = sum(used)
    return {"value": round(len(used) * B / dt, 3), "unit": "clips/s", "cores": threads, "kind": "port",
            "sample": f"BASELINE config #1: {len(used)} eval-mode forwards of batch {B} x 10 s @ 32 kHz waveforms (mel front end + "
                      f"network, no patchout, 1190 tokens, fp32 torch CPU restatement of the reference, {threads} threads of "
                      f"{os.cpu_count()} hw threads), {dt:.1f} s"}


def _sha16(path):
    import hashlib
    with open(path, "rb") as f:
        return hashlib.sha256(f.read()).hexdigest()[:16]


def committed_traffic(name, sources):
    """HBM bytes per launch from a committed PMC profile (the counters cannot be read inside an un-profiled run).  The JSON
    records the sha256 of the kernel sources it was measured on; if they changed since, the number is STALE and is not
    reported (traffic = null and the reason in traffic_source) -- re-run tools/collect_profiles.sh."""
    path = os.path.join(ROOT, "profiles", name)
    if not os.path.isfile(path):
        return None, f"profiles/{name} missing: run tools/collect_profiles.sh"
    with open(path) as f:
        tj = json.load(f)
    now = {os.path.basename(p): _sha16(os.path.join(ROOT, p)) for p in sources}
    if tj.get("source_sha16") != now:
        return None, (f"STALE: profiles/{name} was measured on {tj.get('source_sha16')}, the kernels are now {now}; "
                      "re-run tools/collect_profiles.sh")
    return tj["hbm_bytes_per_launch"], f"profiles/{name} (" + tj.get("method", "rocprofv3 --pmc") + ")"


class AutogradStep:
    """What an UNMODIFIED ex_audioset.py / ex_esc50.py runs per step when models.passt / models.preprocess resolve to
    passt_amd (the drop-in path): the caller's own torch code around ``net(x)`` -- mel_forward (ex_audioset.py:142-153),
    my_mixup + spectrogram / target mixing (:171-183, helpers/mixup.py:5-12), F.binary_cross_entropy_with_logits (:181-186)
    or the CE-mixup of ex_esc50.py:159-165, ``loss.backward()`` through the ONE autograd node of the network,
    torch.optim.AdamW over ``net.parameters()`` (:104-109; Lightning's zero_grad / step order).  Under torch.autocast, as
    Lightning's precision=16 does.  N > 1: passt_amd.ddp.attach(net) -- the node all-reduces per-block buckets from inside
    its backward (no DistributedDataParallel wrapper).  Same .step(x, y) / .reducer / .close() surface as TrainStep."""

    def __init__(self, net, mel, lr, weight_decay, loss, mixup_alpha, precision, comm_dtype, transport, optimizer="adamw", mixup="ref"):
        from passt_amd import ddp
        self.net, self.mel, self.loss, self.alpha = net, mel, loss, mixup_alpha
        if mixup == "pa":
            from passt_amd.mixup import my_mixup                    # the one-word import change (results already on the device)
        else:
            def my_mixup(size, alpha):                              # helpers/mixup.py:5-12 as is: CPU permutation, CPU lam
                rn_indices = torch.randperm(size)
                lambd = np.random.beta(alpha, alpha, size).astype(np.float32)
                lambd = np.concatenate([lambd[:, None], 1 - lambd[:, None]], 1).max(1)
                return rn_indices, torch.FloatTensor(lambd)
        self.my_mixup = my_mixup
        self.autocast = precision == "bf16"
        self.phases = {} if os.environ.get("PASST_AMD_BENCH_PHASES") == "1" else None
        self._mark = None
        net.precision = None                    # follow torch.autocast, like the reference under Lightning AMP
        self.reducer = ddp.attach(net, comm_dtype=comm_dtype, transport=transport)
        if optimizer == "adamw":
            self.opt = torch.optim.AdamW(net.parameters(), lr=lr, weight_decay=weight_decay)       # ex_audioset.py:108, as is
        elif optimizer == "pa_adamw":
            from passt_amd import optim as pa_optim                                                 # the one-word change
            self.opt = pa_optim.AdamW(net.parameters(), lr=lr, weight_decay=weight_decay)
        else:
            self.opt = torch.optim.SGD(net.parameters(), lr=lr)

    def step(self, x, y):
        if self.phases is not None:
            return self._step_phases(x, y)
        return self._step(x, y)

    def _step_phases(self, x, y):
        """PASST_AMD_BENCH_PHASES=1: the same step with a device synchronize between its phases (diagnostic; not a bench mode)"""
        marks = []

        def mark(name):
            torch.cuda.synchronize()
            marks.append((name, time.perf_counter()))
        self._mark = mark
        mark("start")
        out = self._step(x, y)
        mark("optimizer")
        for (_, t0), (n, t1) in zip(marks, marks[1:]):
            self.phases[n] = self.phases.get(n, 0.0) + (t1 - t0)
        self.phases["steps"] = self.phases.get("steps", 0) + 1
        self._mark = None
        return out

    def _step(self, x, y):
        F = torch.nn.functional
        net = self.net
        mark = self._mark or (lambda name: None)
        if self.mel is not None:
            old_shape = x.size()
            x = self.mel(x.reshape(-1, old_shape[2]))
            x = x.reshape(old_shape[0], old_shape[1], x.shape[1], x.shape[2])
        mark("mel")
        B = len(y)
        perm, lam = self.my_mixup(B, self.alpha)                                              # ex_audioset.py:173
        lam = lam.to(x.device)                                                                # :174
        x = x * lam.reshape(B, 1, 1, 1) + x[perm] * (1.0 - lam.reshape(B, 1, 1, 1))          # :175-176
        mark("mixup")
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=self.autocast):
            y_hat, _ = net(x)
            mark("forward")
            if self.loss == "bce":
                y_mix = y * lam.reshape(B, 1) + y[perm] * (1.0 - lam.reshape(B, 1))
                loss = F.binary_cross_entropy_with_logits(y_hat, y_mix, reduction="none").mean()
            else:
                sl = F.cross_entropy(y_hat, y, reduction="none") * lam + F.cross_entropy(y_hat, y[perm], reduction="none") * (1.0 - lam)
                loss = sl.mean()
        mark("loss")
        self.opt.zero_grad()
        loss.backward()
        mark("backward")
        self.opt.step()
        return loss.detach()

    def close(self):
        from passt_amd import ddp
        ddp.detach(self.net)
        if self.phases:
            n = max(self.phases.pop("steps", 1), 1)
            print("autograd step phases (ms, device synchronised between them): "
                  + ", ".join(f"{k} {1e3 * v / n:.3f}" for k, v in self.phases.items()), file=sys.stderr, flush=True)


def speed_test_flow(net, batch_size, warmup=10, test_length=100, compile_net=True, spectrogram="ones"):
    """The reference's ``model_speed_test`` (ex_audioset.py:364-426), statement for statement, around ``net``:
    x = ones(B,1,128,998), target = ones(B,527) (:384-385), GradScaler (:389), ``torch.compile(net)`` (:391),
    SGD(lr=0.001) over the compiled module's parameters (:392), 10 warm-up + 100 timed iterations of
    autocast (fp16, ``torch.cuda.amp.autocast()``'s default) -> net -> BCE mean -> scaler.scale(loss).backward() ->
    scaler.step -> scaler.update (:398-404, :413-419), device-synchronised wall clock (:405-406, :420-421).  No zero_grad --
    the reference has none either: gradients accumulate into .grad across iterations (the autograd node's fresh gradient is
    added to the kept one by AccumulateGrad), which this flow therefore pays for like the reference does.
    Returns specs/second and what the scaler / compiler did."""
    import torch.nn.functional as F
    from torch._dynamo.utils import counters
    dev = next(net.parameters()).device
    x = torch.ones([batch_size, 1, 128, 998], device=dev) if spectrogram == "ones" else torch.randn([batch_size, 1, 128, 998], device=dev)
    target = torch.ones([batch_size, 527], device=dev)
    scaler = torch.cuda.amp.GradScaler()
    counters.clear()
    if compile_net:
        net = torch.compile(net)
    optimizer = torch.optim.SGD(net.parameters(), lr=0.001)

    def one():
        with torch.cuda.amp.autocast():
            y_hat, embed = net(x)
            loss = F.binary_cross_entropy_with_logits(y_hat, target, reduction="none").mean()
        scaler.scale(loss).backward()
        scaler.step(optimizer)
        scaler.update()
        return loss, y_hat

    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        torch.cuda.synchronize()
        t1 = time.time()
        for _ in range(warmup):
            loss, y_hat = one()
        torch.cuda.synchronize()
        t_warm = time.time() - t1
        import gc
        gc.collect()
        gc.freeze()             # torch.compile imported ~900 modules: keep full collections of them out of the timed loop
        frames_after_warmup = counters["frames"].get("total", 0)
        first_loss = float(loss.item())
        t1 = time.time()
        for _ in range(test_length):
            loss, y_hat = one()
        torch.cuda.synchronize()
        t2 = time.time()
    return {"specs_per_second": test_length * batch_size / (t2 - t1), "ms_per_iteration": 1e3 * (t2 - t1) / test_length,
            "warmup_s": t_warm, "first_loss": first_loss, "last_loss": float(loss.item()), "logits_dtype": str(y_hat.dtype),
            "grad_scale": float(scaler.get_scale()),
            "dynamo": {"graphs_captured": counters["stats"].get("unique_graphs", 0), "frames_after_warmup": frames_after_warmup,
                       "frames_total": counters["frames"].get("total", 0)}}


def run_speed_test(args):
    """`bench.py --speedtest`: the reference's own definition of its speed metric, one JSON line per batch size."""
    import passt_amd
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    for B in ([args.batch] if args.batch else [64, 12]):          # BASELINE config #2's batch; the reference's default (:365)
        torch.manual_seed(1234)
        with warnings.catch_warnings(), contextlib.redirect_stdout(sys.stderr):
            warnings.simplefilter("ignore")
            net = passt_amd.get_model(arch=ARCH, pretrained=False, n_classes=527, s_patchout_t=40, s_patchout_f=4).to(dev).train()
        r = speed_test_flow(net, B, warmup=max(args.warmup, 1), test_length=args.steps, compile_net=not args.no_compile)
        gflop = algorithmic_gflop_per_clip()
        print(json.dumps({"metric": "specs/second, model_speed_test (ex_audioset.py:364-426): torch.compile(net) + fp16 autocast + "
                                    "GradScaler + SGD, x = ones(B,1,128,998), target = ones(B,527)",
                          "value": round(r["specs_per_second"], 1), "unit": "specs/s", "n_gpus": 1, "steps": args.steps,
                          "warmup": max(args.warmup, 1), "ms_per_step": round(r["ms_per_iteration"], 3), "higher_is_better": True,
                          "dtype": "bf16 MFMA under the caller's fp16 autocast (f32 logits / gradients)", "data": "synthetic",
                          "config": {"workload": "passt_s (768/12/12, s_patchout_t=40,f=4: 474 tokens), net only (no mel, no mixup), "
                                                 "drop-in autograd path, torch.optim.SGD, no zero_grad (as the reference)",
                                     "per_gpu_batch": B, "compiled": not args.no_compile},
                          "mfma_frac_end_to_end": round(r["specs_per_second"] * gflop / 1e3 / BF16_MFMA_PEAK_TFLOPS, 4),
                          "first_loss": round(r["first_loss"], 6), "last_loss": round(r["last_loss"], 6),
                          "grad_scale_after": r["grad_scale"], "dynamo": r["dynamo"], "logits_dtype": r["logits_dtype"]}), flush=True)
        del net


def error_line(args, msg, **extra):
    """The contract's ONE JSON line when no measurement exists: same keys, value null, the reason in `error`."""
    cfgd = CONFIGS[args.config]
    print(json.dumps({"metric": cfgd["metric"], "value": None, "unit": "clips/s", "n_gpus": args.gpus, "steps": args.steps,
                      "warmup": args.warmup, "ms_per_step": None, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
                      "dtype": "bf16" if args.precision == "bf16" else "f32", "data": "synthetic",
                      "config": {"workload": cfgd["desc"].format(opt=args.optimizer), "baseline_config": args.config},
                      "error": msg, **extra}), flush=True)


def self_launch(args):
    """`python bench.py --gpus N` without a launcher around it: fork one rank per GPU ourselves -- what the reference's own
    entry point does (`DDP=N python ex_audioset.py`, ex_audioset.py:499-524: MASTER_ADDR 127.0.0.1, a random port, one child
    per rank, rank r on device r).  Children are fresh interpreters of this same command line with RANK / LOCAL_RANK /
    WORLD_SIZE / MASTER_* set, i.e. exactly what torch.distributed.run would have started; rank 0's stdout (the ONE JSON
    line) is passed through, the other ranks' stdout goes to stderr.  A rank that dies takes the job down: the survivors
    are killed (they would hang in a collective) and an error line is printed.  Returns the exit code."""
    import socket
    import subprocess
    n = args.gpus
    dry = os.environ.get("PASST_AMD_BENCH_DRY_GLOO") == "1"
    have = torch.cuda.device_count() if torch.cuda.is_available() else 0
    if have < (1 if dry else n):
        error_line(args, f"--gpus {n} needs {n} visible HIP devices, this box has {have}"
                         + (" (PASST_AMD_BENCH_DRY_GLOO=1 runs every rank on device 0 over gloo: needs one)" if dry else ""),
                   devices_visible=have)
        return 2
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    import signal
    import tempfile
    procs = []
    # rank 0's stdout goes to a file, not a pipe: nothing can block on a full pipe while the parent polls.  The children stay
    # in the parent's process group (a group kill by whoever started us takes them along) and are killed if we are terminated.
    out0_file = tempfile.TemporaryFile()

    def reap(signum=None, frame=None):
        for p in procs:
            if p.poll() is None:
                p.kill()
        if signum is not None:
            sys.exit(128 + signum)
    for sig in (signal.SIGTERM, signal.SIGINT, signal.SIGHUP):
        signal.signal(sig, reap)
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), PASST_AMD_BENCH_CHILD="1")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=out0_file if r == 0 else sys.stderr))
    failed = None
    deadline = time.time() + float(os.environ.get("PASST_AMD_BENCH_TIMEOUT_S", "1500"))
    try:
        while True:
            rcs = [p.poll() for p in procs]
            bad = [(r, rc) for r, rc in enumerate(rcs) if rc not in (None, 0)]
            if bad:
                failed = f"rank {bad[0][0]} exited with code {bad[0][1]}"
            elif time.time() > deadline:
                failed = "timed out"
            if failed or all(rc == 0 for rc in rcs):
                break
            time.sleep(0.2)
    finally:
        reap()
    for p in procs:
        p.wait()
    out0_file.seek(0)
    out0 = out0_file.read().decode(errors="replace")
    lines = [l for l in out0.splitlines() if l.startswith("{")]
    if failed or len(lines) != 1:
        error_line(args, failed or f"rank 0 printed {len(lines)} JSON lines", launcher="self (one child per rank)")
        return 1
    print(lines[0], flush=True)
    return 0


def sweep(args):
    """`--sweep-gpus 1,2,4,8`: one child `python bench.py --gpus N ...` per N, in turn (each self-launches its ranks); every child's
    ONE JSON line is passed through as it arrives, so a single command yields the weak-scaling curve (per-GPU batch fixed) with
    the rank / communicator evidence in each N > 1 line.  Efficiency is the reader's to compute (value_N / (N * value_1)).
    Returns non-zero if any N failed (its error line is still printed)."""
    import subprocess
    try:
        ns = [int(t) for t in args.sweep_gpus.split(",") if t.strip()]
        assert ns and all(n >= 1 for n in ns)
    except (ValueError, AssertionError):
        error_line(args, f"--sweep-gpus wants a comma-separated list of GPU counts, got {args.sweep_gpus!r}")
        return 2
    argv, skip = [], False
    for a in sys.argv[1:]:                         # this command line without --sweep-gpus / --gpus
        if skip:
            skip = False
        elif a in ("--sweep-gpus", "--gpus"):
            skip = True
        elif not a.startswith(("--sweep-gpus=", "--gpus=")):
            argv.append(a)
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "MASTER_PORT")}
    worst = 0
    for n in ns:
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--gpus", str(n)] + argv, env=env, stdout=subprocess.PIPE)
        lines = [l for l in r.stdout.decode(errors="replace").splitlines() if l.startswith("{")]
        if len(lines) == 1:
            print(lines[0], flush=True)
        else:
            args.gpus = n
            error_line(args, f"the --gpus {n} child printed {len(lines)} JSON lines (rc {r.returncode})")
        worst = max(worst, abs(r.returncode))
    return worst


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", default="c2", choices=list(CONFIGS), help="BASELINE.json configuration (default c2 = the metric's)")
    ap.add_argument("--batch", type=int, default=0, help="clips per GPU (0 = the configuration's)")
    ap.add_argument("--comm-dtype", default="fp32", choices=["fp32", "bf16"], help="wire format of the gradient all-reduce")
    ap.add_argument("--transport", default="torch", choices=["torch", "rccl_abi"],
                    help="N > 1: torch.distributed (nccl = RCCL) or the library's own pa_comm_* entry points")
    ap.add_argument("--precision", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true", help="skip the per-launch HIP events (measures their cost)")
    ap.add_argument("--no-mel", action="store_true", help="feed spectrograms (reference model_speed_test style)")
    ap.add_argument("--optimizer", default="adamw", choices=["adamw", "sgd", "pa_adamw"],
                    help="adamw: TrainStep's fused pa_adamw resp. (--path autograd) torch.optim.AdamW exactly as ex_audioset.py:108 builds "
                         "it; pa_adamw (autograd path only): passt_amd.optim.AdamW, the same update as one fused launch")
    ap.add_argument("--overlap-wgrad", action="store_true",
                    help="weight-gradient kernels on a second stream, one launch per problem (A/B only: since the batched "
                         "per-block launch it measures the same as the default, profiles/r03_finish_stream_experiment.txt)")
    ap.add_argument("--graph", action="store_true",
                    help="TrainStep(graph=True): forward + loss + backward + per-bucket AdamW captured once as a hipGraph and replayed "
                         "(single GPU; no per-launch events, so the line carries no roofline object: a host-overhead measurement)")
    ap.add_argument("--path", default="trainstep", choices=["trainstep", "autograd"],
                    help="trainstep: passt_amd.train.TrainStep (fused mixup / loss / AdamW, no autograd graph).  autograd: what an "
                         "UNMODIFIED ex_audioset.py runs -- mel -> torch mixup -> net(x) (one autograd Function) -> torch BCE -> "
                         "loss.backward() -> torch.optim.AdamW, gradients all-reduced per block from inside the backward "
                         "(passt_amd.ddp.attach)")
    ap.add_argument("--sweep-gpus", default="",
                    help="comma-separated GPU counts, e.g. 1,2,4,8: runs this same command once per N (self-launching each, one "
                         "after the other) and prints ONE JSON line per N -- the whole weak-scaling curve from one command on an "
                         "8-GPU node; the N = 1 line is the BENCH configuration")
    ap.add_argument("--mixup", default="ref", choices=["ref", "pa"],
                    help="--path autograd: ref = helpers/mixup.py's my_mixup as is (CPU results: three synchronising copies per step); "
                         "pa = passt_amd.mixup.my_mixup, the one-word import change (same draws, results already on the device)")
    ap.add_argument("--speedtest", action="store_true",
                    help="the reference's model_speed_test flow (ex_audioset.py:364-426) around passt_amd's module: torch.compile + "
                         "fp16 autocast + GradScaler + SGD on x = ones(B,1,128,998); one line per batch (64 and 12, or --batch)")
    ap.add_argument("--no-compile", action="store_true", help="--speedtest without torch.compile(net) (A/B)")
    args = ap.parse_args()

    if args.sweep_gpus:
        sys.exit(sweep(args))
    if args.speedtest:
        if not torch.cuda.is_available():
            error_line(args, "no HIP device visible (passt_amd has no CPU path)", devices_visible=0)
            sys.exit(2)
        if args.steps == 20 and args.warmup == 5:
            args.steps, args.warmup = 100, 10                 # the reference's test_length / warm-up (:381, :397)
        run_speed_test(args)
        return
    if "WORLD_SIZE" not in os.environ and "RANK" not in os.environ:
        if args.gpus > 1:
            sys.exit(self_launch(args))
        if not torch.cuda.is_available():
            error_line(args, "no HIP device visible (passt_amd has no CPU path)", devices_visible=0)
            sys.exit(2)
    rank = int(os.environ.get("RANK", "0"))
    try:
        run(args)
    except BaseException as e:      # noqa: BLE001 -- the contract line must exist whatever happened
        if isinstance(e, SystemExit) and e.code in (0, None):
            raise
        # under the self-launcher the parent prints the line (it also sees ranks that were killed)
        if rank == 0 and os.environ.get("PASST_AMD_BENCH_CHILD") != "1":
            error_line(args, f"{type(e).__name__}: {e}"[:500])
        raise


def multi_gpu_diagnostics(ts, x, y, barrier, args, rank, local_rank, world, dev):
    """N > 1, after the timed region: one instrumented step (per-bucket HIP events), the same buckets all-reduced on an idle
    GPU, what the communicator reports, every rank's device.  Returns (allreduce, rccl) on rank 0, (None, None) elsewhere."""
    import torch.distributed as dist
    if os.environ.get("PASST_AMD_BENCH_DIAG_FAIL_RANK") == str(rank):      # test hook (tests/test_gpu_ddp.py): this rank's diagnostics fail
        raise RuntimeError("injected failure of the N > 1 diagnostics on this rank")
    ts.reducer.start_timing()
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        ts.step(x, y)
    barrier()
    buckets = ts.reducer.timing_summary()
    ts.reducer.timing = None
    # the same buckets once more with the GPU otherwise idle: what the wire alone costs (the in-step figure above is
    # launch -> released and therefore includes queueing behind the backward's kernels: it under-reports the bus)
    idle = ts.reducer.measure_idle()
    comm_info = ts.reducer.comm_info()
    props = torch.cuda.get_device_properties(dev)
    mine = {"rank": rank, "device": f"cuda:{local_rank}", "name": props.name, "gcn_arch": getattr(props, "gcnArchName", None),
            "pci_bus_id": getattr(props, "pci_bus_id", None), "comm_nranks": comm_info.get("nranks"), "comm_rank": comm_info.get("rank")}
    devices = [None] * world
    dist.all_gather_object(devices, mine)
    if rank != 0:
        return None, None
    allreduce = {"kind": "MEASURED on this run (one instrumented step after the timed region, rank 0)",
                 "transport": args.transport, "wire_dtype": args.comm_dtype, "buckets": buckets,
                 "bytes_per_step": sum(b["bytes"] for b in buckets),
                 "exposed_wait_ms_per_step": round(sum(b["exposed_wait_ms"] for b in buckets), 4),
                 "bus_GBps_min_max": [min(b["bus_GBps"] for b in buckets), max(b["bus_GBps"] for b in buckets)] if buckets else None,
                 "idle": {"kind": "the same buckets all-reduced with the compute stream idle (best of 3 after a warm-up pass)",
                          "buckets": idle, "ms_per_step": round(sum(b["ms"] for b in idle), 4),
                          "bus_GBps_total": round(2.0 * (world - 1) / world * sum(b["bytes"] for b in idle)
                                                  / max(sum(b["ms"] for b in idle), 1e-6) / 1e6, 1) if idle else None}}
    return allreduce, dict(comm_info, ranks=devices)


def run(args):
    if args.optimizer == "pa_adamw" and args.path != "autograd":
        raise ValueError("--optimizer pa_adamw is the autograd path's fused optimizer; TrainStep's adamw already is the fused kernel")
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # dry run of the N > 1 code path on a box with ONE GPU (tests/test_gpu_ddp.py::test_bench_multi_rank_dry_run): every rank
    # on device 0 over gloo.  The line it prints says so (config.parallelism) and is not a measurement of anything.
    dry = os.environ.get("PASST_AMD_BENCH_DRY_GLOO") == "1"
    if dry:
        local_rank = 0
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world:
        raise RuntimeError(f"--gpus {args.gpus} but the launcher's WORLD_SIZE is {world}")
    if not dry and torch.cuda.device_count() <= local_rank:
        raise RuntimeError(f"rank {rank}: LOCAL_RANK {local_rank} but only {torch.cuda.device_count()} HIP devices are visible")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    import torch.distributed as dist
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        # the all-reduce kernels run on the backend's communication streams: high priority, so that a bucket launched from the
        # backward starts at the next kernel boundary instead of queueing behind the compute stream's launches
        os.environ.setdefault("TORCH_NCCL_HIGH_PRIORITY", "1")
        if dry:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    import passt_amd
    from passt_amd import ops
    from passt_amd.train import TrainStep

    cfgd = CONFIGS[args.config]
    torch.manual_seed(1234 + rank)
    np.random.seed(1234 + rank)
    # module construction prints the reference's notices ("Warning: FMAX is None ..."): keep stdout for the ONE JSON line
    with warnings.catch_warnings(), contextlib.redirect_stdout(sys.stderr):
        warnings.simplefilter("ignore")
        if cfgd["custom"] is not None:
            net = passt_amd.PaSST(img_size=(128, 998), stride=10, num_classes=cfgd["n_classes"], distilled=True, **cfgd["custom"])
        else:
            net = passt_amd.get_model(arch=cfgd["arch"], pretrained=False, n_classes=cfgd["n_classes"], **cfgd["net_kw"])
        net = net.to(dev).train()
        mel = None if args.no_mel else passt_amd.AugmentMelSTFT(
            n_mels=128, sr=32000, win_length=800, hopsize=320, n_fft=1024, fmin=0.0, fmax=None,
            fmin_aug_range=10, fmax_aug_range=2000, **cfgd["mel_kw"]).to(dev).train()      # ex_audioset.py:66-69 / ex_esc50.py:62-66
    net.precision = args.precision
    net.overlap_wgrad = args.overlap_wgrad
    # TrainStep / ddp.attach broadcast rank 0's parameters themselves (identical replicas)
    if args.path == "autograd":
        ts = AutogradStep(net, mel, lr=2e-5, weight_decay=1e-4, loss=cfgd["loss"], mixup_alpha=0.3, precision=args.precision,
                          comm_dtype=args.comm_dtype, transport=args.transport, optimizer=args.optimizer, mixup=args.mixup)
    else:
        ts = TrainStep(net, mel, lr=2e-5, weight_decay=1e-4, optimizer=args.optimizer, mixup_alpha=0.3, use_mixup=True,
                       loss=cfgd["loss"], comm_dtype=args.comm_dtype, transport=args.transport, graph=args.graph)
    B = args.batch or cfgd["batch"]
    frames = 998 if cfgd["clip"] == CLIP_SAMPLES else 1 + (cfgd["clip"] - 1) // 320     # --no-mel: the reference's speed-test shape
    if args.no_mel:
        x = torch.randn(B, 1, 128, frames, device=dev)
    else:
        x = (torch.rand(B, 1, cfgd["clip"], device=dev) * 2 - 1) * 0.1    # U(-1,1)*0.1
    if cfgd["loss"] == "bce":
        y = (torch.rand(B, cfgd["n_classes"], device=dev) < 2.7 / 527).float()      # ~2.7 labels per clip
    else:
        y = torch.randint(0, cfgd["n_classes"], (B,), device=dev)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for _ in range(args.warmup):
            ts.step(x, y)
        barrier()
        # the per-launch HIP events of the instrumented steps are thousands of small Python objects: keep the cyclic collector
        # from walking everything the process has imported while the clock runs (objects alive now become permanent)
        import gc
        gc.collect()
        gc.freeze()
        if getattr(ts, "phases", None) is not None:
            ts.phases.clear()                   # phase diagnostics: timed steps only (warm-up carries one-time module loads)
        # per-launch HIP events on the GEMM family (roofline): two event records per launch cost several % of the step
        # when every step is instrumented, so one timed step in PROFILE_EVERY carries them (profiled_steps)
        prof = {} if (rank == 0 and not args.no_roofline and not args.graph) else None
        prof_at = set(profiled_steps(args.steps))
        t0 = time.perf_counter()
        for i in range(args.steps):
            ops.GEMM_PROFILE = prof if (prof is not None and i in prof_at) else None
            loss = ts.step(x, y)
        ops.GEMM_PROFILE = None
        barrier()
        t1 = time.perf_counter()
    elapsed = torch.tensor([t1 - t0], device=dev, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(elapsed, op=dist.ReduceOp.MAX)
    elapsed = float(elapsed.item())
    loss_v = float(loss.item())
    assert np.isfinite(loss_v), "non-finite loss"

    # ---- outside the timed region -------------------------------------------------------------------------------
    # (a) N > 1: MEASURED per-bucket all-reduce (HIP events: launch -> released, and how long the compute stream was
    #     blocked on it) of one extra instrumented step, next to the modelled curve.  These diagnostics are more collectives
    #     after the measurement is already complete: whatever happens in them (an exception on some rank, a peer that is
    #     gone, a collective that never returns) must not cost the line -- they run under a watchdog, and a rank that fails
    #     leaves without entering another collective (DIAG_TIMEOUT_S later the others print / exit without the diagnostics).
    allreduce = rccl = None
    diag = {"error": None, "done": False}
    report_lock = threading.Lock()

    # (b) the parity mode (exact-f32 MFMA, <= 3e-6 of the fp32 reference: the bound north_star states) trains this fast
    parity_clips = None
    if world == 1 and args.config == "c2" and args.precision == "bf16" and not args.no_cpu_baseline and args.path == "trainstep":
        net.precision = "fp32"
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            ts.step(x, y)
            torch.cuda.synchronize()
            tp0 = time.perf_counter()
            for _ in range(3):
                ts.step(x, y)
            torch.cuda.synchronize()
            parity_clips = round(3 * B / (time.perf_counter() - tp0), 1)
        net.precision = args.precision

    printed = {"done": False}

    def report(allreduce, rccl):
        """rank 0: the ONE JSON line (once: from the normal path, or from the watchdog if the N > 1 diagnostics hang)"""
        with report_lock:
            if printed["done"]:
                return
            printed["done"] = True
            clips = world * B * args.steps
            value = clips / elapsed
            Dm, depth, _ = cfgd["dims"]
            gflop_clip = algorithmic_gflop_per_clip(cfgd["tokens"], Dm, depth, cfgd["kept"])
            out = {
                "metric": cfgd["metric"], "value": round(value, 2), "unit": "clips/s",
                "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                "ms_per_step": round(1e3 * elapsed / args.steps, 3), "higher_is_better": True, "scaling": "weak",
                "vs_baseline": None, "dtype": "bf16" if args.precision == "bf16" else "f32", "data": "synthetic",
                **({"allreduce_bus_GBps_idle": allreduce["idle"]["bus_GBps_total"],
                    "allreduce_exposed_wait_ms_per_step": allreduce["exposed_wait_ms_per_step"],
                    "rccl_nranks": rccl.get("nranks")} if allreduce is not None else {}),
                "config": {"workload": cfgd["desc"].format(opt=args.optimizer), "baseline_config": args.config,
                           "global_batch": world * B, "per_gpu_batch": B, "parallelism": f"dp{world}" + (" DRY RUN: all ranks on one device over gloo" if dry else ""),
                           "grad_wire_dtype": args.comm_dtype if world > 1 else None,
                           "path": ("TrainStep (explicit kernel sequence, fused mixup / loss / AdamW + weight staging)"
                                    + (", network forward + loss + backward + AdamW replayed from one captured hipGraph" if args.graph else "")
                                    if args.path == "trainstep" else
                                    "autograd drop-in: torch mixup -> net(x) -> torch loss -> loss.backward() -> torch.optim."
                                    + {"adamw": "AdamW (multi-tensor default)", "sgd": "SGD", "pa_adamw": "AdamW replaced by passt_amd.optim.AdamW"}[args.optimizer]
                                    + (", my_mixup replaced by passt_amd.mixup.my_mixup" if args.mixup == "pa" else "")
                                    + (", passt_amd.ddp.attach(net)" if world > 1 else "")),
                           "gemm_launch": ("one work item per workgroup (PA_GEMM_NO_PERSIST: the all-reduce kernels share the CUs)"
                                           if getattr(net, "_gemm_flags", 0) & ops._lib.GEMM_NO_PERSIST else "persistent, 256 workgroups"),
                           "input": f"spectrogram (B,1,128,{frames})" if args.no_mel else f"waveform (B,1,{cfgd['clip']}) f32 resident in HBM",
                       "cache_policy": "non-temporal by role in the GEMM / optimizer kernels (DESIGN 4.1, profiles/r06_cache_policy.txt); "
                                       "a library built with -DPA_NO_CACHE_POLICY runs the default policy"},
                "algorithmic_gflop_per_clip": round(gflop_clip, 2),
                "mfma_frac_end_to_end": round(value / world * gflop_clip / 1e3 / BF16_MFMA_PEAK_TFLOPS, 4),
                # the same on the FLOPs actually executed: the prefix-only tail skips part of the last block (exactly)
                "mfma_frac_end_to_end_executed": round(value / world * (gflop_clip - skipped_tail_gflop_per_clip(cfgd["tokens"], Dm)) / 1e3
                                                       / BF16_MFMA_PEAK_TFLOPS, 4),
                "loss": round(loss_v, 6),
            }
            if prof:
                n_prof_steps = len(profiled_steps(args.steps))
                side = {}
                for kind in ("attn_fwd", "attn_bwd", "mel"):
                    recs = prof.pop(kind, None)
                    if recs:
                        ms = sum(s.elapsed_time(e) for s, e, _ in recs)
                        side[kind] = (len(recs), ms, sum(w for _, _, w in recs))
                tot_ms, tot_flop, n, per_kind = 0.0, 0.0, 0, {}
                for kind, recs in prof.items():
                    ms = sum(s.elapsed_time(e) for s, e, _ in recs)
                    fl = sum(f for _, _, f in recs)
                    per_kind[kind] = {"launches": len(recs), "avg_us": round(1e3 * ms / len(recs), 2),
                                      "tflops": round(fl / ms / 1e9, 1)}
                    tot_ms += ms
                    tot_flop += fl
                    n += len(recs)
                achieved = tot_flop / tot_ms / 1e9
                ms_step = 1e3 * elapsed / args.steps
                # north_star: "MFMA utilisation for attention/MLP": the attention kernels, timed the same way in the same run
                if "attn_fwd" in side and "attn_bwd" in side:
                    (nf, msf, wf), (nb, msb, wb) = side["attn_fwd"], side["attn_bwd"]
                    out["attention"] = {"bound": "mfma (co-bound by VALU: one exp per score, head dim 64)",
                                        "fwd_avg_us": round(1e3 * msf / nf, 2), "fwd_tflops": round(wf / msf / 1e9, 1),
                                        "bwd_avg_us": round(1e3 * msb / nb, 2), "bwd_tflops": round(wb / msb / 1e9, 1),
                                        "achieved": round((wf + wb) / (msf + msb) / 1e9, 1), "peak": BF16_MFMA_PEAK_TFLOPS,
                                        "unit": "TFLOP/s", "frac": round((wf + wb) / (msf + msb) / 1e9 / BF16_MFMA_PEAK_TFLOPS, 4),
                                        "flops": "algorithmic: 4 N^2 64 per head forward, 10 N^2 64 backward (5 products); "
                                                 "last block: 2 queries only",
                                        "time_share_of_step": round((msf + msb) / n_prof_steps / ms_step, 4)}
                # north_star: "rocprof-reported HBM GB/s for the front end": live number here, PMC traffic in profiles/
                if "mel" in side:
                    nm, msm, wm = side["mel"]
                    mt, mt_src = (committed_traffic("r06_mel_traffic.json", ["passt_amd/csrc/mel.hip"])
                                  if (args.config == "c2" and B == 64) else (None, "only collected for config c2, B = 64"))
                    out["frontend"] = {"bound": "hbm", "kernel": "pa::mel_frontend_kernel (STFT + mel + log + SpecAugment, one launch)",
                                       "avg_us": round(1e3 * msm / nm, 2), "achieved": round(wm / msm / 1e6, 1), "peak": HBM_PEAK_GBPS,
                                       "unit": "GB/s", "frac": round(wm / msm / 1e6 / HBM_PEAK_GBPS, 4),
                                       "algorithmic_bytes_per_launch": round(wm / nm), "traffic": mt, "traffic_source": mt_src,
                                       "time_share_of_step": round(msm / n_prof_steps / ms_step, 4)}
                # HBM traffic per launch of the same kernel family: PMC counters cannot be read from inside this
                # process; they are collected with `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` (separate passes)
                # over this very command and committed (tools/gemm_traffic.py -> profiles/r01_gemm_traffic.json)
                traffic, traffic_src = (committed_traffic("r06_gemm_traffic.json", ["passt_amd/csrc/gemm.hip"])
                                        if (args.config == "c2" and B == 64 and args.precision == "bf16")
                                        else (None, "only collected for config c2, B = 64, bf16"))
                out["roofline"] = {"bound": "mfma", "achieved": round(achieved, 1), "peak": BF16_MFMA_PEAK_TFLOPS,
                                   "unit": "TFLOP/s", "frac": round(achieved / BF16_MFMA_PEAK_TFLOPS, 4), "traffic": traffic,
                                   "traffic_unit": "bytes per launch", "traffic_source": traffic_src,
                                   "algorithmic_flop_per_launch": round(tot_flop / n),
                                   "kernel": "GEMM family pa::gemm_nt_stagger_kernel / gemm_nt_kernel / gemm_tn_stagger_kernel <bf16> "
                                             "(all epilogues + weight gradients; 2*M*N*K algorithmic FLOPs per launch)",
                                   "launches": n, "avg_launch_us": round(1e3 * tot_ms / n, 2),
                                   "sampling": f"HIP events around every GEMM launch of 1 timed step in {PROFILE_EVERY} (steps {', '.join(map(str, list(profiled_steps(args.steps))[:3]))}{', ...' if len(profiled_steps(args.steps)) > 3 else ''})",
                                   "gemm_time_share_of_step": round(tot_ms / n_prof_steps / ms_step, 4),
                                   "per_epilogue": per_kind}
            # multi-GPU: SCALE is measured by the driver when it has an 8-GPU node; what can be said from ONE GPU is a model
            bus_eff, bus_src = 0.35, "35 % (an ASSUMED RCCL bus efficiency: no N > 1 measurement in this run)"
            idle_bus = (allreduce or {}).get("idle", {}).get("bus_GBps_total") if not dry else None
            if idle_bus:
                bus_eff = idle_bus / (min(world - 1, 7) * 153.0)
                bus_src = (f"{100 * bus_eff:.1f} % = this run's own idle all-reduce of the same buckets ({idle_bus} GB/s bus bandwidth at "
                           f"N = {world}, allreduce_measured.idle) over the link peak")
            out["scaling_model"] = {"kind": "MODELLED, not measured" + (" (wire term from this run's measured bus bandwidth)" if idle_bus else ""),
                                    "inputs": "this run's ms_per_step, the reducer's bucket bytes (one bucket per block, launched from the "
                                    f"backward), min(N-1, 7) xGMI links x 153 GB/s at {bus_src}, GEMM slow-down next to a co-running "
                                    "whole-CU kernel from profiles/r04_copersist_probe.txt (worst case: +28 %)",
                                    "bus_efficiency": round(bus_eff, 4),
                                    "bucket_MB": {str(k): round(v / 1e6, 2) for k, v in ts.reducer.bucket_bytes().items()},
                                    "n_gpus": modelled_scaling(1e3 * elapsed / args.steps, ts.reducer.bucket_bytes(), bus_eff=bus_eff)}
            if world == 1 and not args.no_cpu_baseline and args.config == "c2":
                out["cpu_baseline"] = cpu_baseline()
                out["cpu_baseline_forward"] = cpu_baseline_forward()
            if parity_clips is not None:
                out["parity_mode_clips_s"] = parity_clips     # same step, precision="fp32": what meets the <= 1e-3 parity bound
            if allreduce is not None:
                out["allreduce_measured"] = allreduce
                out["rccl"] = rccl
            if diag["error"]:
                out["multi_gpu_diagnostics_error"] = diag["error"]
            print(json.dumps(out), flush=True)

    def watchdog_fire():
        if diag["done"]:
            return
        diag["error"] = (f"the N > 1 diagnostics after the timed region did not finish within {DIAG_TIMEOUT_S} s on rank {rank}; "
                         "value / ms_per_step are the completed measurement")
        if rank == 0:
            try:
                torch.cuda.set_device(local_rank)
                report(None, None)
            finally:
                os._exit(0)
        os._exit(0)

    watchdog = None
    if world > 1:
        watchdog = threading.Timer(DIAG_TIMEOUT_S if rank == 0 else DIAG_TIMEOUT_S + 20, watchdog_fire)
        watchdog.daemon = True
        watchdog.start()
        try:
            allreduce, rccl = multi_gpu_diagnostics(ts, x, y, barrier, args, rank, local_rank, world, dev)
        except Exception as e:      # noqa: BLE001 -- diagnostics only
            diag["error"] = f"{type(e).__name__}: {e}"[:300]
            allreduce = rccl = None
            if rank != 0:
                # no further collective on this rank (the peers would wait for it in vain): leave; rank 0's watchdog prints
                print(f"[bench rank {rank}] N > 1 diagnostics failed: {diag['error']}", file=sys.stderr, flush=True)
                os._exit(0)
    diag["done"] = True
    if watchdog is not None:
        watchdog.cancel()
    if rank == 0:
        report(allreduce, rccl)
    if world > 1 and diag["error"]:
        # the ranks are no longer in step with each other: no closing barrier / communicator teardown (they could hang)
        sys.stdout.flush()
        os._exit(0)
    ts.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
