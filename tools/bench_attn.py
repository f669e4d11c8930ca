#!/usr/bin/env python
"""Attention kernels alone (HIP events, random data) at the shapes of BASELINE configs #2 / #5.

    PASST_AMD_LIB=passt_amd/libpasst_amd_x.so python tools/bench_attn.py [--tag x] [--shapes 64x12x474,12x12x353]

One JSON line per shape: forward / backward us, TF/s on the algorithmic 4 N^2 64 (fwd) and 10 N^2 64 (bwd, five
products) FLOP counts, fraction of the 2.5 PF/s bf16 MFMA peak.  Run once per library build for A/B (same box, one
gpurun call).
"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from passt_amd import ops  # noqa: E402


def timeit(fn, iters):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = []
    for _ in range(5):
        a.record()
        for _ in range(iters):
            fn()
        b.record()
        torch.cuda.synchronize()
        best.append(a.elapsed_time(b) / iters * 1e-3)
    return sorted(best)[len(best) // 2]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tag", default=os.path.basename(os.environ.get("PASST_AMD_LIB", "libpasst_amd.so")))
    ap.add_argument("--shapes", default="64x12x474,12x12x353,64x12x1190")
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--dtype", default="bf16")
    args = ap.parse_args()
    dt = torch.bfloat16 if args.dtype == "bf16" else torch.float32
    for shp in args.shapes.split(","):
        B, H, N = (int(v) for v in shp.split("x"))
        D = H * 64
        g = torch.Generator(device="cuda").manual_seed(1)
        qkv = (torch.randn(B * N, 3 * D, device="cuda", generator=g)).to(dt)
        d_o = torch.randn(B * N, D, device="cuda", generator=g).to(dt)
        o, lse = ops.attention_fwd(qkv, B, H, N, 0.125, flags=ops.ATTN_Q_PRESCALED)
        tf = timeit(lambda: ops.attention_fwd(qkv, B, H, N, 0.125, flags=ops.ATTN_Q_PRESCALED), args.iters)
        tb = timeit(lambda: ops.attention_bwd(qkv, o, d_o, lse, B, H, N, 0.125, flags=ops.ATTN_Q_PRESCALED), args.iters)
        ff, fb = 4.0 * N * N * 64 * B * H, 10.0 * N * N * 64 * B * H
        print(json.dumps({"lib": args.tag, "shape": shp, "dtype": args.dtype, "fwd_us": round(tf * 1e6, 1),
                          "bwd_us": round(tb * 1e6, 1), "fwd_tflops": round(ff / tf / 1e12, 1),
                          "bwd_tflops": round(fb / tb / 1e12, 1), "fwd_frac": round(ff / tf / 2.5e15, 4),
                          "bwd_frac": round(fb / tb / 2.5e15, 4)}), flush=True)


if __name__ == "__main__":
    main()
