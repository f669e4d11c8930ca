#!/usr/bin/env python
"""Where a front-end workgroup spends its life: the probe build of mel.hip (make mel_variant NAME=probe MEL_DEFS=-DPA_MEL_PROBE)
leaves s_memrealtime stamps (100 MHz) of every wave's phases in the output tile.
    PASST_AMD_LIB=passt_amd/libpasst_amd_mel_probe.so python tools/probe_mel.py"""
import json
import os
import sys
import warnings

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import passt_amd  # noqa: E402

PHASES = ["start", "table loads issued, span staged", "geometry + band-stage constants + 2 barriers", "4 frames", "barrier", "epilogue"]
NS = len(PHASES)

with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    mel = passt_amd.AugmentMelSTFT(fmin_aug_range=10, fmax_aug_range=2000).to("cuda").eval()
B, L = 64, 320000
wave = (torch.rand(B, L, device="cuda") * 2 - 1) * 0.1
for _ in range(5):
    out = mel(wave)
torch.cuda.synchronize()
o = out.cpu().numpy()                                  # [B][128][T]
T = o.shape[2]
f0 = np.arange(0, T, 16)
f0 = f0[f0 + 16 <= T]                                  # full tiles only
st = o[:, :64, f0].reshape(B, 4, 16, len(f0))            # [b][wave][stamp][wg]
start = st[:, :, 0]                                    # 24-bit absolute
d = st[:, :, 1:NS]                                     # deltas from stamp 0, 10 ns units
seg = np.diff(np.concatenate([np.zeros_like(d[:, :, :1]), d], axis=2), axis=2) * 0.01     # us per phase
res = {"unit": "us", "workgroups": int(B * len(f0)), "phases": {}}
for i, name in enumerate(PHASES[1:]):
    v = seg[:, :, i].reshape(-1)
    res["phases"][name] = {"median": round(float(np.median(v)), 2), "p10": round(float(np.percentile(v, 10)), 2),
                           "p90": round(float(np.percentile(v, 90)), 2)}
life = d[:, :, NS - 2].reshape(-1) * 0.01
res["wave_lifetime"] = {"median": round(float(np.median(life)), 2), "p10": round(float(np.percentile(life, 10)), 2),
                        "p90": round(float(np.percentile(life, 90)), 2)}
s0 = start[:, 0].reshape(-1)
s0 = (s0 - s0.min()) % (1 << 24)
res["launch_span_us"] = round(float((s0 + d[:, 0, NS - 2].reshape(-1)).max() * 0.01), 1)
hist, _ = np.histogram(s0 * 0.01, bins=12, range=(0, 120))
res["workgroup_starts_per_10us"] = hist.tolist()
print(json.dumps(res, indent=1), flush=True)
