#!/usr/bin/env python
"""Single-pass attention backward against the two-kernel backward on the same inputs, per output third (dq / dk / dv), with
the positions of the worst entries -- localises an indexing mistake (dk / dv: phase 1; dq: the T transposition / phase 2).

    python tools/debug_attn_fused.py [N ...]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from passt_amd import ops  # noqa: E402


def main():
    Ns = [int(a) for a in sys.argv[1:]] or [32, 64, 20, 67, 474, 512]
    for N in Ns:
        B, H = 2, 2
        D = H * 64
        g = torch.Generator(device="cuda").manual_seed(N)
        qkv = torch.randn(B * N, 3 * D, device="cuda", generator=g).to(torch.bfloat16)
        d_o = torch.randn(B * N, D, device="cuda", generator=g).to(torch.bfloat16)
        o, lse = ops.attention_fwd(qkv, B, H, N, 0.125, flags=ops.ATTN_Q_PRESCALED)
        ref = ops.attention_bwd(qkv, o, d_o, lse, B, H, N, 0.125, flags=ops.ATTN_Q_PRESCALED | ops.ATTN_BWD_TWO_PASS).float()
        got = ops.attention_bwd(qkv, o, d_o, lse, B, H, N, 0.125, flags=ops.ATTN_Q_PRESCALED | ops.ATTN_BWD_SINGLE_PASS).float()
        torch.cuda.synchronize()
        line = [f"N={N}"]
        for j, name in enumerate(("dq", "dk", "dv")):
            a, b = got[:, j * D:(j + 1) * D], ref[:, j * D:(j + 1) * D]
            err = (a - b).abs()
            e = float(err.max() / b.abs().max())
            line.append(f"{name} rel {e:.2e} finite {bool(torch.isfinite(a).all())}")
            if e > 2e-2:
                idx = torch.nonzero(err > 0.02 * b.abs().max())[:6].tolist()
                line.append(f"bad@{idx} got {[round(float(a[i, k]), 3) for i, k in idx[:3]]} want {[round(float(b[i, k]), 3) for i, k in idx[:3]]}")
                bad_rows = sorted(set(i for i, _ in torch.nonzero(err > 0.02 * b.abs().max()).tolist()))
                line.append(f"bad rows {bad_rows[:12]}..{len(bad_rows)} bad cols {sorted(set(k for _, k in torch.nonzero(err > 0.02 * b.abs().max()).tolist()))[:16]}")
        print("  ".join(line), flush=True)


if __name__ == "__main__":
    main()
