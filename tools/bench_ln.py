#!/usr/bin/env python
"""LayerNorm forward / backward of one passt_s block (M = 30336 rows x 768) timed alone with HIP events, GB/s on the
algorithmic bytes; one JSON line (A/B builds with PASST_AMD_LIB=other.so)."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from passt_amd import ops  # noqa: E402
from passt_amd._lib import PA_BF16  # noqa: E402
from bench_kernels import timeit  # noqa: E402


def main():
    M, D = 64 * 474, int(os.environ.get("D", "768"))
    x = torch.randn(M, D, device="cuda")
    g, b = torch.rand(D, device="cuda") + 0.5, torch.randn(D, device="cuda")
    y, mean, rstd = ops.layernorm_fwd(x, g, b, 1e-6, PA_BF16)
    dy = torch.randn(M, D, device="cuda").to(torch.bfloat16)
    dres = torch.randn(M, D, device="cuda")
    dg, db, dc = (torch.zeros(D, device="cuda") for _ in range(3))
    t_f = timeit(lambda: ops.layernorm_fwd(x, g, b, 1e-6, PA_BF16), 30)
    t_b = timeit(lambda: ops.layernorm_bwd(dy, x, g, mean, rstd, dres, dg, db, True, dcolsum=dc), 30)
    el = M * D
    print(json.dumps({"lib": os.environ.get("PASST_AMD_LIB", "default"), "ln_fwd_us": round(t_f * 1e6, 1),
                      "ln_fwd_TBs": round(el * 6 / t_f / 1e12, 2), "ln_bwd_us": round(t_b * 1e6, 1),
                      "ln_bwd_TBs": round(el * 16 / t_b / 1e12, 2)}), flush=True)


if __name__ == "__main__":
    main()
