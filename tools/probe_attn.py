#!/usr/bin/env python
"""Phase timeline of the plain attention forward loop (probe library: make attn_variant NAME=probe
ATTN_DEFS="-DPA_ATTN_PROBE -DPA_ATTN_PIPE=0"; PASST_AMD_LIB=passt_amd/libpasst_amd_attn_probe.so python tools/probe_attn.py).

Every wave stamps s_memtime at the boundaries of a key tile's phases; the kernel sums the differences over all waves and
tiles.  Printed: mean shader cycles per wave-tile spent in  wait+barrier+DMA issue | Q K^T issue | MFMA drain + row max |
reference move + exp | P V issue.
"""
import ctypes as C
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from passt_amd import _lib, ops  # noqa: E402


def main():
    import numpy as np
    lib = _lib.load()
    lib.pa_attn_probe_read.restype = C.c_int
    lib.pa_attn_probe_read.argtypes = [C.c_void_p]
    NW = 16384
    out = np.zeros(NW * 8, dtype=np.uint64)
    B, H, N = 64, 12, 474
    qkv = torch.randn(B * N, 3 * H * 64, device="cuda").to(torch.bfloat16)
    for _ in range(3):
        ops.attention_fwd(qkv, B, H, N, 0.125)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    ops.attention_fwd(qkv, B, H, N, 0.125)
    b.record()
    torch.cuda.synchronize()
    lib.pa_attn_probe_read(out.ctypes.data)
    w = out.reshape(NW, 8)[:3072 * 4].astype(np.float64)
    raw5 = out.reshape(NW, 8)[:3072 * 4, 5]
    hwid = (raw5 >> np.uint64(8)) & np.uint64(0xffffffff)
    xcc = (raw5 >> np.uint64(40)) & np.uint64(0xf)
    w[:, 5] = (raw5 & np.uint64(0xff)).astype(np.float64)
    tiles = w[:, 5].sum()
    names = ["wait_barrier_stage", "qk_issue", "drain_max", "rebase_exp", "pv_issue"]
    res = {n: round(float(w[:, i].sum() / tiles), 1) for i, n in enumerate(names)}
    res["sum"] = round(float(w[:, :5].sum() / tiles), 1)
    life = w[:, 7] - w[:, 6]
    res["wave_lifetime_cycles_mean"] = round(float(life.mean()), 0)
    ok = w[:, 6] > 0
    # s_memtime is per XCD: all time arithmetic inside one XCD
    simd = (hwid >> np.uint64(4)) & np.uint64(3); cu = (hwid >> np.uint64(8)) & np.uint64(15); se = (hwid >> np.uint64(13)) & np.uint64(7)
    sh = (hwid >> np.uint64(12)) & np.uint64(1)
    key = (((xcc * np.uint64(8) + se) * np.uint64(2) + sh) * np.uint64(16) + cu) * np.uint64(4) + simd
    res["distinct_simds"] = int(len(np.unique(key[ok])))
    res["wave_slots_seen"] = sorted(set(int(v) for v in (hwid[ok] & np.uint64(15))))
    cukey = key >> np.uint64(2)
    spans, conc = [], []
    for c in np.unique(cukey[ok])[:64]:
        m = ok & (cukey == c)
        t0 = w[m, 6].min(); span = float(w[m, 7].max() - t0)
        spans.append(span); conc.append(float(life[m].sum() / span / 4))
    res["per_cu_span_cycles_mean"] = round(float(np.mean(spans)), 0)
    res["per_cu_span_cycles_minmax"] = [min(spans), max(spans)]
    res["per_cu_mean_waves_per_simd"] = round(float(np.mean(conc)), 2)
    res["xcc_values"] = sorted(set(int(v) for v in xcc[ok]))
    res["launch_us_events"] = round(a.elapsed_time(b) * 1e3, 1)
    # concurrency: how many waves are alive at the median time
    print(json.dumps(res))


if __name__ == "__main__":
    main()
