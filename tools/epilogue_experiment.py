#!/usr/bin/env python
"""Is the K=768 GEMM epilogue bound by HBM (global) or by per-CU latency?  Time the same 256x256-tile GEMM with
60 / 252 / 504 / 1428 tiles and K = 64 / 768 / 3072; a per-CU-bound epilogue costs the same with 60 tiles as with 252."""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from passt_amd import ops
from passt_amd._lib import EPI_DGELU, EPI_GELU, EPI_RESID, EPI_STORE, PA_BF16
from bench_kernels import timeit

bf = torch.bfloat16
def rnd(*s, dtype=bf):
    return (torch.rand(*s, device="cuda") * 2 - 1).to(dtype)

out = []
for epi, nm in ((EPI_STORE, "store"), (EPI_GELU, "gelu"), (EPI_DGELU, "dgelu")):
    for K in (64, 768, 3072):
        for rows in (5, 21, 42, 119):
            M, Nn = 256 * rows, 3072
            A, W = rnd(M, K), rnd(Nn, K) * 0.05
            bias = torch.zeros(Nn, device="cuda")
            if epi == EPI_STORE:
                kw = dict(bias=bias, out_lp=torch.empty(M, Nn, device="cuda", dtype=bf))
            elif epi == EPI_GELU:
                kw = dict(bias=bias, out_lp=torch.empty(M, Nn, device="cuda", dtype=bf), out_lp2=torch.empty(M, Nn, device="cuda", dtype=bf))
            else:
                kw = dict(aux=rnd(M, Nn), out_lp=torch.empty(M, Nn, device="cuda", dtype=bf))
            ops.GEMM_TUNE = 6   # 256x256 role-split
            sec = timeit(lambda: ops.gemm_nt(A, W, PA_BF16, epi, **kw), 20)
            r = dict(epi=nm, K=K, tiles=rows * 12, us=round(sec * 1e6, 2))
            out.append(r); print(json.dumps(r), flush=True)
ops.GEMM_TUNE = 0
json.dump(out, open("gpurun_out/epilogue_experiment.json", "w"), indent=1)
