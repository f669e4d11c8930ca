#!/usr/bin/env python
"""The fused mel front end alone: 64 x 10 s clips (train-mode module, eval-mode call) timed with HIP events."""
import json
import os
import sys
import warnings

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import passt_amd  # noqa: E402
from bench_kernels import timeit  # noqa: E402

with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    mel = passt_amd.AugmentMelSTFT(fmin_aug_range=10, fmax_aug_range=2000).to("cuda").eval()
out = {"lib": os.environ.get("PASST_AMD_LIB", "default")}
for B, L in ((64, 320000), (96, 160000), (12, 160000)):
    wave = (torch.rand(B, L, device="cuda") * 2 - 1) * 0.1
    sec = timeit(lambda: mel(wave), 30)
    byts = 4.0 * (B * L + B * 128 * (1 + (L - 1) // 320))
    out[f"B{B}_L{L}"] = {"us": round(sec * 1e6, 1), "GBps": round(byts / sec / 1e9, 1), "frac_hbm_peak": round(byts / sec / 8e12, 4)}
print(json.dumps(out), flush=True)
