#!/usr/bin/env python
"""Experiment: does splitting a short-K GEMM into two concurrent launches with different tile heights
(so the per-CU rounds of the two launches drift apart) hide the epilogue write bursts?"""
import sys, os, json
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from passt_amd import ops
from passt_amd._lib import EPI_DGELU, EPI_GELU, EPI_RESID, EPI_STORE, PA_BF16

DEV = "cuda"
M, D = 64 * 474, 768
bf = torch.bfloat16
rnd = lambda *s, dtype=bf: (torch.rand(*s, device=DEV) * 2 - 1).to(dtype)
x = rnd(M, D)
side = torch.cuda.Stream()


def timeit(fn, iters=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(iters):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    ts.sort()
    return ts[len(ts) // 2] * 1e3


for name, N, epi in (("qkv store", 3 * D, EPI_STORE), ("fc1 gelu", 4 * D, EPI_GELU), ("dgelu", 4 * D, EPI_DGELU)):
    W = rnd(N, D) * 0.05
    bias = torch.zeros(N, device=DEV)
    o1 = torch.empty(M, N, device=DEV, dtype=bf)
    o2 = torch.empty(M, N, device=DEV, dtype=bf)
    aux = rnd(M, N)

    def call(rows0, rows1, tune):
        ops.GEMM_TUNE = tune
        kw = dict(out_lp=o1[rows0:rows1])
        if epi == EPI_STORE:
            kw.update(bias=bias)
        elif epi == EPI_GELU:
            kw.update(bias=bias, out_lp2=o2[rows0:rows1])
        else:
            kw.update(aux=aux[rows0:rows1])
        ops.gemm_nt(x[rows0:rows1], W, PA_BF16, epi, **kw)
        ops.GEMM_TUNE = 0

    base = timeit(lambda: call(0, M, 7))
    base6 = timeit(lambda: call(0, M, 6))
    res = {"kernel": name, "single_v7_us": round(base, 1), "single_v6_us": round(base6, 1)}
    for frac in (0.35, 0.5, 0.65):
        m1 = int(M * frac) // 768 * 768          # multiple of both 192 and 256

        def split():
            side.wait_stream(torch.cuda.current_stream())
            call(0, m1, 6)
            with torch.cuda.stream(side):
                call(m1, M, 7)
            torch.cuda.current_stream().wait_stream(side)
        res[f"split_{frac}_us"] = round(timeit(split), 1)

    def split_same():
        m1 = (M // 2) // 768 * 768
        side.wait_stream(torch.cuda.current_stream())
        call(0, m1, 7)
        with torch.cuda.stream(side):
            call(m1, M, 7)
        torch.cuda.current_stream().wait_stream(side)
    res["split_same_variant_us"] = round(timeit(split_same), 1)
    print(json.dumps(res), flush=True)
