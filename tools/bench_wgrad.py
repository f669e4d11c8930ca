#!/usr/bin/env python
"""The batched weight-gradient launch of one passt_s block (pa_gemm_tn_batched: qkv, proj, fc1, fc2 at M = 30336)
and the attention kernels, timed alone with HIP events.  A/B two builds with PASST_AMD_LIB=<other .so>;
run under `rocprofv3 --pmc FETCH_SIZE ...` for the HBM traffic of the same launches."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from passt_amd import ops  # noqa: E402
from passt_amd._lib import PA_BF16  # noqa: E402
from bench_kernels import timeit  # noqa: E402

bf = torch.bfloat16


def rnd(*s, dtype=bf):
    return (torch.rand(*s, device="cuda") * 2 - 1).to(dtype)


def main():
    iters = int(os.environ.get("ITERS", "20"))
    B, N, D, H = 64, 474, 768, 12
    M = B * N
    x, h, q3 = rnd(M, D), rnd(M, 4 * D), rnd(M, 3 * D)
    probs = [(q3, x, torch.empty(3 * D, D, device="cuda"), False), (x, rnd(M, D), torch.empty(D, D, device="cuda"), False),
             (h, x, torch.empty(4 * D, D, device="cuda"), False), (x, h, torch.empty(D, 4 * D, device="cuda"), False)]
    ws = [None]

    def run():
        ws[0] = ops.wgrad_tn_batched(probs, PA_BF16, ws[0])
    out = {"lib": os.environ.get("PASST_AMD_LIB", "default")}
    flops = sum(2.0 * M * p[0].shape[1] * p[1].shape[1] for p in probs)
    sec = timeit(run, iters)
    out["wgrad_batched_us"] = round(sec * 1e6, 1)          # includes the batched split-K reduction
    out["wgrad_batched_tflops"] = round(flops / sec / 1e12, 1)
    if not os.environ.get("NO_ATTN"):
        o, lse = ops.attention_fwd(q3, B, H, N, 0.125)
        do = rnd(M, D)
        sec = timeit(lambda: ops.attention_fwd(q3, B, H, N, 0.125), iters)
        out["attn_fwd_us"] = round(sec * 1e6, 1)
        out["attn_fwd_tflops"] = round(4.0 * N * N * 64 * B * H / sec / 1e12, 1)
        sec = timeit(lambda: ops.attention_bwd(q3, o, do, lse, B, H, N, 0.125), iters)
        out["attn_bwd_us"] = round(sec * 1e6, 1)
        out["attn_bwd_tflops"] = round(10.0 * N * N * 64 * B * H / sec / 1e12, 1)
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
