#!/usr/bin/env python
"""LayerNorm forward / backward alone at the training shapes (HIP events), for A/B builds of layernorm.hip:
    PASST_AMD_LIB=passt_amd/libpasst_amd_ln_<name>.so PA_LN_BWD_BLOCKS=768 python tools/bench_ln.py"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from passt_amd import ops  # noqa: E402
from passt_amd._lib import PA_BF16  # noqa: E402
from bench_kernels import timeit  # noqa: E402

out = {"lib": os.environ.get("PASST_AMD_LIB", "default"), "blocks": os.environ.get("PA_LN_BWD_BLOCKS", "default")}
D = 768
for M in (30336, 4236):
    xf = torch.randn(M, D, device="cuda")
    dres = torch.randn(M, D, device="cuda")
    g, b_ = torch.randn(D, device="cuda"), torch.randn(D, device="cuda")
    y, mean, rstd = ops.layernorm_fwd(xf, g, b_, 1e-6, PA_BF16)
    dy = torch.randn(M, D, device="cuda").to(torch.bfloat16)
    dg, dbt = torch.zeros(D, device="cuda"), torch.zeros(D, device="cuda")
    f = timeit(lambda: ops.layernorm_fwd(xf, g, b_, 1e-6, PA_BF16), 50)
    bw = timeit(lambda: ops.layernorm_bwd(dy, xf, g, mean, rstd, dres, dg, dbt, True, defer=[]), 50)
    out[f"M{M}"] = {"fwd_us": round(f * 1e6, 1), "fwd_TBps": round(M * D * 6 / f / 1e12, 2),
                    "bwd_us": round(bw * 1e6, 1), "bwd_TBps": round(M * D * 16 / bw / 1e12, 2)}
print(json.dumps(out), flush=True)
