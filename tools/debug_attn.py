"""Stress check of the attention kernels at the bench shape: non-finite outputs and run-to-run determinism (the kernels
have no atomics: two launches on the same input must be bit-identical).  Round 3 found a sporadic garbage-row bug this way."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from passt_amd import ops
B, H, N = int(sys.argv[1]) if len(sys.argv) > 1 else 64, 12, 474
D = H * 64
torch.manual_seed(0)
qkv = torch.randn(B * N, 3 * D, device="cuda").to(torch.bfloat16)
d_o = torch.randn(B * N, D, device="cuda").to(torch.bfloat16)
bad = 0
for flags in (0, 1):
    ref = None
    for rep in range(6):
        o, lse = ops.attention_fwd(qkv, B, H, N, 0.125, flags=flags)
        dq = ops.attention_bwd(qkv, o, d_o, lse, B, H, N, 0.125, flags=flags)
        nf = int((~torch.isfinite(o.float())).sum()) + int((~torch.isfinite(dq.float())).sum()) + int((~torch.isfinite(lse)).sum())
        cur = (o.clone(), lse.clone(), dq.clone())
        diff = 0 if ref is None else sum(int((a != b).sum()) for a, b in zip(ref, cur))
        ref = ref or cur
        print(f"flags {flags} rep {rep}: non-finite {nf}, elements differing from rep 0: {diff}")
        bad += nf + diff
print("STRESS", "OK" if bad == 0 else f"FAILED ({bad})")
sys.exit(1 if bad else 0)
