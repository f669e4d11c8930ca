#!/usr/bin/env python
"""The seven forward / input-gradient GEMMs of one passt_s block (M = 30336) timed alone with HIP events; one compact
JSON line (us per GEMM + their sum) so that builds / environment knobs can be A/B'd in one gpurun call:
    PA_NT_STAGGER=50 python tools/bench_nt.py        PASST_AMD_LIB=other.so python tools/bench_nt.py"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from passt_amd import ops  # noqa: E402
from passt_amd._lib import EPI_DGELU, EPI_GELU, EPI_RESID, EPI_STORE, PA_BF16  # noqa: E402
from bench_kernels import timeit  # noqa: E402

bf = torch.bfloat16


def rnd(*s, dtype=bf):
    return (torch.rand(*s, device="cuda") * 2 - 1).to(dtype)


def main():
    iters = int(os.environ.get("ITERS", "20"))
    B = int(os.environ.get("BATCH", "64"))
    M, D = B * 474, 768
    x, h, xf = rnd(M, D), rnd(M, 4 * D), rnd(M, D, dtype=torch.float32)
    out = {"tag": os.environ.get("TAG", ""), "stagger": os.environ.get("PA_NT_STAGGER", "0")}
    tot = 0.0
    for name, A, Nn, K, epi in (("qkv", x, 3 * D, D, EPI_STORE), ("proj", x, D, D, EPI_RESID), ("fc1", x, 4 * D, D, EPI_GELU),
                                ("fc2", h, D, 4 * D, EPI_RESID), ("dfc2", x, 4 * D, D, EPI_DGELU), ("dfc1", h, D, 4 * D, EPI_STORE),
                                ("dqkv", rnd(M, 3 * D), D, 3 * D, EPI_STORE), ("dproj", x, D, D, EPI_STORE)):
        W = rnd(Nn, K) * 0.05
        bias = torch.zeros(Nn, device="cuda")
        if epi == EPI_STORE:
            kw = dict(bias=bias, out_lp=torch.empty(M, Nn, device="cuda", dtype=bf))
        elif epi == EPI_GELU:
            kw = dict(bias=bias, out_lp=torch.empty(M, Nn, device="cuda", dtype=bf),
                      out_lp2=torch.empty(M, Nn, device="cuda", dtype=bf))
        elif epi == EPI_RESID:
            kw = dict(bias=bias, resid=xf, out_f32=torch.empty(M, Nn, device="cuda"))
        else:
            kw = dict(aux=rnd(M, Nn), out_lp=torch.empty(M, Nn, device="cuda", dtype=bf))
        sec = timeit(lambda: ops.gemm_nt(A, W, PA_BF16, epi, **kw), iters)
        out[name] = round(sec * 1e6, 1)
        tot += sec
    out["sum_us"] = round(tot * 1e6, 1)
    out["tflops"] = round(2.0 * M * D * D * (3 + 1 + 4 + 4 + 4 + 4 + 3 + 1) / tot / 1e12, 1)
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
