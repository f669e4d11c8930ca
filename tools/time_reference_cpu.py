#!/usr/bin/env python
"""Time the REAL reference (kkoutini/PaSST imported read-only from /root/reference through oracle/ref_import.py) on this host's
CPU cores, next to the oracle (the port bench.py's cpu_baseline times on the GPU box, where /root/reference does not exist):
BASELINE config #1 (eval forward, mel front end + network, batch 2, no patchout, fp32) and the train-mode fwd + bwd of config #2's
network at batch 2.  Build container only.   python tools/time_reference_cpu.py > profiles/rNN_reference_cpu_timing.txt"""
import os
import sys
import time
import warnings

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import detgen, ref_import      # noqa: E402
from oracle import passt_oracle as O       # noqa: E402


def timed(fn, budget_s=25.0, max_iters=12):
    ts = []
    t_start = time.time()
    while True:
        t0 = time.time()
        fn()
        ts.append(time.time() - t0)
        if time.time() - t_start > budget_s or len(ts) >= max_iters:
            break
    used = ts[1:] if len(ts) > 1 else ts
    return len(used), sum(used)


def main():
    assert ref_import.reference_available(), "needs /root/reference"
    threads = min(32, os.cpu_count() or 1)
    torch.set_num_threads(threads)
    warnings.simplefilter("ignore")
    B = 2
    print(f"host: {os.cpu_count()} hardware threads, torch threads {threads}, torch {torch.__version__}")
    # ---- config #1: eval forward, mel + network, no patchout
    cfg = O.make_cfg()
    sd_np = detgen.passt_state_dict(cfg, 1)
    ref = ref_import.build_reference_passt(cfg, sd_np).eval()
    _, ref_pre = ref_import.load_reference()
    mel = ref_import.run_silently(ref_pre.AugmentMelSTFT, fmin_aug_range=10, fmax_aug_range=2000).eval()
    wave = torch.from_numpy(detgen.uniform(2, "wave", (B, 320000), -0.1, 0.1))
    sd = O.to_torch(sd_np)

    def ref_fwd():
        with torch.no_grad():
            x = mel(wave)
            ref_import.run_silently(ref, x[:, None, :, :998])

    def ora_fwd():
        with torch.no_grad():
            x = O.mel_frontend(wave, training=False, fmin_aug_range=10, fmax_aug_range=2000)
            O.passt_forward(sd, x[:, None, :, :998], cfg, training=False)
    for name, fn in (("REFERENCE", ref_fwd), ("oracle   ", ora_fwd)):
        n, dt = timed(fn)
        print(f"config #1 eval forward (mel + net, B = {B}, N = 1190, fp32)   {name}: {n * B / dt:6.3f} clips/s  ({n} iterations, {dt:.1f} s)")
    # ---- config #2 network: train-mode fwd + bwd, s_patchout t = 40 / f = 4
    cfg = O.make_cfg(s_patchout_t=40, s_patchout_f=4)
    sd_np = detgen.passt_state_dict(cfg, 1)
    ref = ref_import.build_reference_passt(cfg, sd_np).train()
    sd = O.to_torch(sd_np, requires_grad=True)
    x = torch.from_numpy(detgen.uniform(1, "x", (B, 1, 128, 998), -1, 1))
    y = (torch.rand(B, 527) < 0.005).float()

    def ref_train():
        ref.zero_grad()
        lo, _ = ref_import.run_silently(ref, x)
        torch.nn.functional.binary_cross_entropy_with_logits(lo, y, reduction="none").mean().backward()

    def ora_train():
        for p in sd.values():
            p.grad = None
        lo, _ = O.passt_forward(sd, x, cfg, training=True)
        O.bce_loss(lo, y).backward()
    for name, fn in (("REFERENCE", ref_train), ("oracle   ", ora_train)):
        n, dt = timed(fn)
        print(f"config #2 network train fwd + bwd (B = {B}, N = 474, fp32)        {name}: {n * B / dt:6.3f} clips/s  ({n} iterations, {dt:.1f} s)")


if __name__ == "__main__":
    main()
