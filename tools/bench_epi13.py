#!/usr/bin/env python
"""Every reachable tile variant (pa_gemm_args.tune) of the two MLP-epilogue GEMMs of a passt_s block at the benchmarked batch
-- fc1 + GELU (PA_EPI_GELU) and dgrad-fc2 x GELU' (PA_EPI_DGELU), M = 30 336, N = 3072, K = 768, bf16 -- timed alone with HIP
events, row-major and (where the variant has it) with the pre-activation in the library's blocked layout; results checked
against the default variant (bit-equal activations / gradients are expected: same products, same k order).

    python tools/bench_epi13.py > gpurun_out/r06_gemm_variants_epi13.txt

VERDICT r5 item 3: the table pick_nt_variant's rates are re-fitted from, and the measurement of the two-workgroups-per-CU tiles
(tunes 3 / 9) with the blocked pre-activation.  Reference op: Mlp.forward, models/passt.py:284-290."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from passt_amd import ops  # noqa: E402
from passt_amd._lib import EPI_DGELU, EPI_GELU, GEMM_BLOCKED_PRE, PA_BF16  # noqa: E402
from bench_kernels import timeit  # noqa: E402

bf = torch.bfloat16
NAMES = {0: "library choice", 1: "128x128 4 waves 2-stage (2+ WG/CU)", 2: "256x256 8 waves lockstep", 3: "192x128 4 waves, 80 KiB: 2 WG/CU",
         9: "128x256 4 waves 3x24 KiB: 2 WG/CU", 6: "256x256 role-split", 7: "192x256 role-split", 8: "128x256 role-split",
         17: "192x256 role-split, A 2 tiles ahead", 18: "128x256 role-split, A 2 tiles ahead"}
BLOCKED = (0, 3, 9, 6, 7, 8, 17, 18)


def main():
    iters = int(os.environ.get("ITERS", "30"))
    B = int(os.environ.get("BATCH", "64"))
    M, D, N = B * 474, 768, 3072
    g = torch.Generator(device="cuda").manual_seed(3)
    x = (torch.rand(M, D, device="cuda", generator=g) * 2 - 1).to(bf)
    W = ((torch.rand(N, D, device="cuda", generator=g) * 2 - 1) * 0.05).to(bf)
    bias = (torch.rand(N, device="cuda", generator=g) - 0.5)
    dy = (torch.rand(M, D, device="cuda", generator=g) * 2 - 1).to(bf)
    Wt = ((torch.rand(N, D, device="cuda", generator=g) * 2 - 1) * 0.05).to(bf)      # fc2.weight^T: [3072][768]
    flops = 2.0 * M * N * D
    nelem = ops._lib.load().pa_gemm_blocked_pre_elems(M, N)
    ref = {}
    print(f"# M = {M}, N = {N}, K = {D}, bf16, median of {iters} launches alone (HIP events); TF/s = 2MNK / time; frac of 2.5 PF/s")
    print("# epilogue tune layout us TF/s frac  max|diff| vs the library choice  (variant)")
    for layout in ("blocked", "rowmajor"):
        for tune in (0, 17, 7, 6, 18, 8, 9, 3, 2, 1):
            if layout == "blocked" and tune not in BLOCKED:
                continue
            fl = GEMM_BLOCKED_PRE if layout == "blocked" else 0
            pre = torch.zeros(nelem if layout == "blocked" else M * N, device="cuda", dtype=bf).view(-1, N)
            act = torch.empty(M, N, device="cuda", dtype=bf)
            try:
                sec = timeit(lambda: ops.gemm_nt(x, W, PA_BF16, EPI_GELU, bias=bias, out_lp=pre, out_lp2=act, flags=fl, tune=tune, M=M, N=N, K=D), iters)
            except Exception as e:      # noqa: BLE001
                print(f"gelu {tune} {layout} unsupported ({str(e)[:60]})")
                continue
            key = ("gelu", layout)
            if tune == 0:
                ref[key] = (pre.clone(), act.clone())
            d = max(float((pre.float() - ref[key][0].float()).abs().max()), float((act.float() - ref[key][1].float()).abs().max()))
            print(f"gelu {tune} {layout} {sec * 1e6:.1f} {flops / sec / 1e12:.0f} {flops / sec / 2.5e15:.3f}  {d:.3g}  ({NAMES[tune]})", flush=True)
            # GELU' with the pre-activation this variant's fc1 wrote (the blocked layout is the same for every tile height)
            dpre = torch.empty(M, N, device="cuda", dtype=bf)
            cs_out = torch.empty(N, device="cuda")
            cws = ops.gemm_colsum_ws(M, N, "cuda")
            # one call with the finishing reduction (values), then timed as the step issues it: column sums left as partial rows
            ops.gemm_nt(dy, Wt, PA_BF16, EPI_DGELU, aux=pre, out_lp=dpre, colsum_out=cs_out, colsum_ws=cws, flags=fl, tune=tune, M=M, N=N, K=D)
            sec = timeit(lambda: ops.gemm_nt(dy, Wt, PA_BF16, EPI_DGELU, aux=pre, out_lp=dpre, colsum_out=cs_out, colsum_ws=cws,
                                             flags=fl | ops._lib.GEMM_COLSUM_DEFER, tune=tune, M=M, N=N, K=D), iters)
            key = ("dgelu", layout)
            if tune == 0:
                ref[key] = (dpre.clone(), cs_out.clone())
            d = float((dpre.float() - ref[key][0].float()).abs().max())
            dc = float((cs_out - ref[key][1]).abs().max() / (ref[key][1].abs().max() + 1e-30))
            print(f"dgelu {tune} {layout} {sec * 1e6:.1f} {flops / sec / 1e12:.0f} {flops / sec / 2.5e15:.3f}  {d:.3g} (bias-sum rel {dc:.2g})  ({NAMES[tune]})", flush=True)


if __name__ == "__main__":
    main()
