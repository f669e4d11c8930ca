#!/usr/bin/env python
"""ESC-50 shapes (M = B * 353 rows): every NT tile variant (pa_gemm_args.tune) on the block's GEMM shapes, unsplit, plus the
split-K partial launch at 2..6 slices for the [M][768] outputs -- the table pick_nt_variant / pa_gemm_nt_splitk_plan are
calibrated against.   python tools/sweep_nt_small.py [B]"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from passt_amd import ops  # noqa: E402
from passt_amd._lib import EPI_GELU, EPI_PARTIAL, EPI_RESID, EPI_STORE, PA_BF16  # noqa: E402
from bench_kernels import timeit  # noqa: E402

bf = torch.bfloat16
B = int(sys.argv[1]) if len(sys.argv) > 1 else 12
M = B * 353


def rnd(*s, dtype=bf):
    return (torch.rand(*s, device="cuda") * 2 - 1).to(dtype)


out = {"M": M}
for name, N, K, epi in (("fc2_resid", 768, 3072, EPI_RESID), ("dfc1_store", 768, 3072, EPI_STORE), ("dqkv_store", 768, 2304, EPI_STORE),
                        ("proj_resid", 768, 768, EPI_RESID), ("qkv_store", 2304, 768, EPI_STORE), ("fc1_gelu", 3072, 768, EPI_GELU)):
    A, W = rnd(M, K), rnd(N, K) * 0.05
    bias = torch.zeros(N, device="cuda")
    if epi == EPI_STORE:
        kw = dict(bias=bias, out_lp=torch.empty(M, N, device="cuda", dtype=bf))
    elif epi == EPI_GELU:
        kw = dict(bias=bias, out_lp=torch.empty(M, N, device="cuda", dtype=bf), out_lp2=torch.empty(M, N, device="cuda", dtype=bf))
    else:
        kw = dict(bias=bias, resid=rnd(M, N, dtype=torch.float32), out_f32=torch.empty(M, N, device="cuda"))
    row = {}
    ops.GEMM_TUNE = 0
    row["default"] = round(timeit(lambda: ops.gemm_nt(A, W, PA_BF16, epi, **kw), 30) * 1e6, 1)
    for tune in (1, 2, 3, 9, 6, 7, 8, 17, 18):
        ops.GEMM_TUNE = tune
        try:
            row[f"t{tune}"] = round(timeit(lambda: ops.gemm_nt(A, W, PA_BF16, epi, **kw), 30) * 1e6, 1)
        except Exception as e:  # noqa: BLE001
            row[f"t{tune}"] = str(e)[:30]
    if N == 768 and K >= 2304:
        for tune in (8, 18, 7, 17):
            for sk in (2, 3, 4, 6):
                ops.GEMM_TUNE = tune
                part = torch.empty(sk, M, N, device="cuda")
                try:
                    row[f"partial_t{tune}_s{sk}"] = round(timeit(lambda: ops.gemm_nt(A, W, PA_BF16, EPI_PARTIAL, out_f32=part, split_k=sk), 30) * 1e6, 1)
                except Exception as e:  # noqa: BLE001
                    row[f"partial_t{tune}_s{sk}"] = str(e)[:30]
    ops.GEMM_TUNE = 0
    out[name] = row
    print(name, json.dumps(row), flush=True)
