#!/usr/bin/env python
"""Context for the roofline fractions: this library's kernels next to the vendor path on the same shapes, same box.

    python tools/bench_vs_vendor.py > gpurun_out/vs_vendor.json

* NT GEMM + bias (PA_EPI_STORE)      vs  torch.nn.functional.linear (bf16: hipBLASLt / rocBLAS behind torch)
* weight gradient dY^T X (split-K, f32 result, incl. the reduction)  vs  torch.matmul(dY.T, X) (bf16 result)
* attention forward / backward       vs  torch scaled_dot_product_attention (whatever ROCm backend torch picks)

HIP events, median of 5 batches of `iters` launches, operands resident (both sides see the same cache state).  The vendor
numbers are a yardstick, not a component: nothing in passt_amd calls these torch ops."""
import json
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from passt_amd import ops  # noqa: E402
from passt_amd._lib import EPI_STORE, PA_BF16  # noqa: E402


def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(5):
        a.record()
        for _ in range(iters):
            fn()
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) / iters * 1e-3)
    return sorted(ts)[2]


def main():
    dev = "cuda"
    g = torch.Generator(device=dev).manual_seed(3)
    rows = []
    M = 64 * 474
    for name, N, K in [("qkv", 2304, 768), ("proj / dgrad-proj", 768, 768), ("fc1 (no GELU)", 3072, 768),
                       ("fc2 / dgrad-fc1", 768, 3072), ("dgrad-qkv", 768, 2304)]:
        x = (torch.rand(M, K, device=dev, generator=g) - 0.5).bfloat16()
        w = (torch.rand(N, K, device=dev, generator=g) - 0.5).bfloat16()
        b = torch.rand(N, device=dev, generator=g)
        bb = b.bfloat16()
        out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        t_ours = timeit(lambda: ops.gemm_nt(x, w, PA_BF16, EPI_STORE, bias=b, out_lp=out))
        t_vend = timeit(lambda: F.linear(x, w, bb))
        fl = 2.0 * M * N * K
        rows.append({"kernel": f"NT GEMM + bias, {name}", "M": M, "N": N, "K": K, "ours_us": round(t_ours * 1e6, 1),
                     "vendor_us": round(t_vend * 1e6, 1), "ours_tflops": round(fl / t_ours / 1e12, 1),
                     "vendor_tflops": round(fl / t_vend / 1e12, 1)})
    # the four weight gradients of a block in one launch vs four vendor GEMMs
    probs, fl = [], 0.0
    for N, K in [(768, 3072), (3072, 768), (768, 768), (2304, 768)]:
        dY = (torch.rand(M, N, device=dev, generator=g) - 0.5).bfloat16()
        X = (torch.rand(M, K, device=dev, generator=g) - 0.5).bfloat16()
        probs.append((dY, X, torch.empty(N, K, device=dev), False, None))
        fl += 2.0 * M * N * K
    state = {"ws": None}

    def ours_wgrad():
        state["ws"] = ops.wgrad_tn_batched(probs, PA_BF16, state["ws"])

    t_ours = timeit(ours_wgrad, 10)
    t_vend = timeit(lambda: [torch.matmul(dY.t(), X) for dY, X, _, _, _ in probs], 10)
    rows.append({"kernel": "weight gradients of one block (4 problems; ours: one launch + reduction, f32 result; vendor: 4 GEMMs, bf16 result)",
                 "ours_us": round(t_ours * 1e6, 1), "vendor_us": round(t_vend * 1e6, 1),
                 "ours_tflops": round(fl / t_ours / 1e12, 1), "vendor_tflops": round(fl / t_vend / 1e12, 1)})
    # attention
    for B, H, N in [(64, 12, 474), (12, 12, 353)]:
        D = H * 64
        qkv = torch.randn(B * N, 3 * D, device=dev, generator=g).bfloat16()
        d_o = torch.randn(B * N, D, device=dev, generator=g).bfloat16()
        o, lse = ops.attention_fwd(qkv, B, H, N, 0.125)
        tf = timeit(lambda: ops.attention_fwd(qkv, B, H, N, 0.125))
        tb = timeit(lambda: ops.attention_bwd(qkv, o, d_o, lse, B, H, N, 0.125))
        q, k, v = (t.contiguous().requires_grad_(True) for t in qkv.view(B, N, 3, H, 64).permute(2, 0, 3, 1, 4))
        do4 = d_o.view(B, N, H, 64).permute(0, 2, 1, 3).contiguous()
        tvf = timeit(lambda: F.scaled_dot_product_attention(q.detach(), k.detach(), v.detach()))
        ov = F.scaled_dot_product_attention(q, k, v)

        def vend_bwd():
            q.grad = k.grad = v.grad = None
            ov.backward(do4, retain_graph=True)

        tvb = timeit(vend_bwd)
        ff, fb = 4.0 * N * N * 64 * B * H, 10.0 * N * N * 64 * B * H
        rows.append({"kernel": f"attention B={B} H={H} N={N} d=64", "ours_fwd_us": round(tf * 1e6, 1), "vendor_fwd_us": round(tvf * 1e6, 1),
                     "ours_bwd_us": round(tb * 1e6, 1), "vendor_bwd_us": round(tvb * 1e6, 1),
                     "ours_fwd_tflops": round(ff / tf / 1e12, 1), "vendor_fwd_tflops": round(ff / tvf / 1e12, 1),
                     "ours_bwd_tflops": round(fb / tb / 1e12, 1), "vendor_bwd_tflops": round(fb / tvb / 1e12, 1)})
    print(json.dumps({"torch": torch.__version__, "device": torch.cuda.get_device_name(0), "rows": rows}, indent=1))


if __name__ == "__main__":
    main()
