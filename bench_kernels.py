#!/usr/bin/env python
"""Isolated-kernel benchmark at the BASELINE configs[1] shapes (B=64, N=474, M=30336, D=768, H=12):
every hot kernel timed alone with HIP events on random data, reported against its roofline
(bf16 MFMA 2.5 PF dense for GEMM/attention, HBM 8 TB/s for the streaming kernels).

    python bench_kernels.py [--iters 20] [--out gpurun_out/kernels.json]
"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from passt_amd import ops  # noqa: E402
from passt_amd._lib import EPI_DGELU, EPI_GELU, EPI_RESID, EPI_STORE, PA_BF16, PA_F32  # noqa: E402

DEV = "cuda"
MFMA_PEAK, HBM_PEAK = 2500.0, 8000.0


def timeit(fn, iters):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
    for a, b in evs:
        a.record()
        fn()
        b.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in evs)
    return ts[len(ts) // 2] * 1e-3     # median seconds


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--out", default="gpurun_out/kernels.json")
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--variants", default="0", help="comma list of pa_gemm_args.tune values to time the NT GEMMs with")
    args = ap.parse_args()
    B, N, D, H = args.batch, 474, 768, 12
    M = B * N
    bf = torch.bfloat16
    res = []

    def rnd(*s, dtype=bf):
        return (torch.rand(*s, device=DEV) * 2 - 1).to(dtype)

    def add(name, sec, flops=None, bytes_=None):
        r = {"kernel": name, "us": round(sec * 1e6, 2)}
        if flops:
            r["tflops"] = round(flops / sec / 1e12, 1)
            r["frac_mfma_peak"] = round(flops / sec / 1e12 / MFMA_PEAK, 4)
        if bytes_:
            r["gbps"] = round(bytes_ / sec / 1e9, 1)
            r["frac_hbm_peak"] = round(bytes_ / sec / 1e9 / HBM_PEAK, 4)
        res.append(r)
        print(json.dumps(r), flush=True)

    x = rnd(M, D)
    h = rnd(M, 4 * D)
    xf = rnd(M, D, dtype=torch.float32)
    # ---- forward / dgrad GEMMs (NT) ----
    for name, A, Nn, K, epi in (("gemm qkv (store)", x, 3 * D, D, EPI_STORE), ("gemm proj (resid)", x, D, D, EPI_RESID),
                                ("gemm fc1 (gelu)", x, 4 * D, D, EPI_GELU), ("gemm fc2 (resid)", h, D, 4 * D, EPI_RESID),
                                ("gemm dgrad fc2 (dgelu)", x, 4 * D, D, EPI_DGELU),
                                ("gemm dgrad fc1 (store)", h, D, 4 * D, EPI_STORE),
                                ("gemm dgrad qkv (store)", rnd(M, 3 * D), D, 3 * D, EPI_STORE)):
        W = rnd(Nn, K) * 0.05
        bias = torch.zeros(Nn, device=DEV)
        kw = {}
        if epi == EPI_STORE:
            kw = dict(bias=bias, out_lp=torch.empty(M, Nn, device=DEV, dtype=bf))
        elif epi == EPI_GELU:
            kw = dict(bias=bias, out_lp=torch.empty(M, Nn, device=DEV, dtype=bf),
                      out_lp2=torch.empty(M, Nn, device=DEV, dtype=bf))
        elif epi == EPI_RESID:
            kw = dict(bias=bias, resid=xf, out_f32=torch.empty(M, Nn, device=DEV))
        elif epi == EPI_DGELU:
            kw = dict(aux=rnd(M, Nn), out_lp=torch.empty(M, Nn, device=DEV, dtype=bf))
        for var in [int(v) for v in args.variants.split(",")]:
            ops.GEMM_TUNE = var
            sec = timeit(lambda: ops.gemm_nt(A, W, PA_BF16, epi, **kw), args.iters)
            add(name + f" M{M} N{Nn} K{K} [variant {var}]", sec, flops=2.0 * M * Nn * K)
        ops.GEMM_TUNE = 0
    # ---- weight-gradient GEMMs (TN, in place) ----
    for name, dY, X in (("wgrad qkv", rnd(M, 3 * D), x), ("wgrad proj", x, x), ("wgrad fc1", h, x), ("wgrad fc2", x, h)):
        dW = torch.empty(dY.shape[1], X.shape[1], device=DEV)
        ws = [None]

        def run():
            ws[0] = ops.wgrad_tn(dY, X, dW, PA_BF16, False, ws[0])
        for var, label in ((1, "128x128"), (0, "256x256 role-split")):
            ops.GEMM_TUNE = var
            sec = timeit(run, args.iters)
            add(name + f" (tn {label}, split-K+reduce) N{dY.shape[1]} K{X.shape[1]} M{M}", sec,
                flops=2.0 * M * dY.shape[1] * X.shape[1])
        ops.GEMM_TUNE = 0
    db = torch.empty(4 * D, device=DEV)
    sec = timeit(lambda: ops.colsum(h, db), args.iters)
    add("colsum (bias grad) [M,3072] bf16", sec, bytes_=M * 4 * D * 2)
    # ---- attention ----
    qkv = rnd(M, 3 * D)
    o, lse = ops.attention_fwd(qkv, B, H, N, 0.125, flags=ops.ATTN_Q_PRESCALED)
    sec = timeit(lambda: ops.attention_fwd(qkv, B, H, N, 0.125, flags=ops.ATTN_Q_PRESCALED), args.iters)
    add(f"attention fwd B{B} H{H} N{N}", sec, flops=4.0 * N * N * 64 * B * H)
    do = rnd(M, D)
    sec = timeit(lambda: ops.attention_bwd(qkv, o, do, lse, B, H, N, 0.125, flags=ops.ATTN_Q_PRESCALED), args.iters)
    add(f"attention bwd (library's choice: single pass for B*H >= 512, N <= 512) B{B} H{H} N{N}", sec, flops=10.0 * N * N * 64 * B * H)
    sec = timeit(lambda: ops.attention_bwd(qkv, o, do, lse, B, H, N, 0.125, flags=ops.ATTN_Q_PRESCALED | ops.ATTN_BWD_TWO_PASS), args.iters)
    add(f"attention bwd, two kernels (dq + dkdv) B{B} H{H} N{N}", sec, flops=10.0 * N * N * 64 * B * H)
    # ---- layer norm ----
    g, b_ = torch.ones(D, device=DEV), torch.zeros(D, device=DEV)
    y, mean, rstd = ops.layernorm_fwd(xf, g, b_, 1e-6, PA_BF16)
    sec = timeit(lambda: ops.layernorm_fwd(xf, g, b_, 1e-6, PA_BF16), args.iters)
    add("layernorm fwd [M,768] f32->bf16", sec, bytes_=M * D * 6)
    dg, dbt = torch.empty(D, device=DEV), torch.empty(D, device=DEV)
    sec = timeit(lambda: ops.layernorm_bwd(x, xf, g, mean, rstd, xf, dg, dbt, True), args.iters)
    add("layernorm bwd [M,768] (dy bf16, x/dres f32 -> dx f32 + bf16)", sec, bytes_=M * D * (2 + 4 + 4 + 4 + 2))
    # ---- front end ----
    import warnings

    import passt_amd
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        mel = passt_amd.AugmentMelSTFT(fmin_aug_range=10, fmax_aug_range=2000).to(DEV).eval()
    wave = (torch.rand(B, 320000, device=DEV) * 2 - 1) * 0.1
    sec = timeit(lambda: mel(wave), args.iters)
    add(f"mel front end B{B} x 10 s (algorithmic 1.792 MB/clip)", sec, bytes_=B * 1.792e6)
    os.makedirs(os.path.dirname(args.out) or ".", exist_ok=True)
    with open(args.out, "w") as f:
        json.dump(res, f, indent=1)


if __name__ == "__main__":
    main()
