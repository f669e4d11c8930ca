#!/usr/bin/env python
"""Barrier-interval timeline of the role-split weight-gradient kernel (probe library:
PASST_AMD_LIB=passt_amd/libpasst_amd_probe.so python tools/probe_tn.py).

For stages 4..11 of the first work item of workgroups 0..7 the probe build stamps s_memtime at four points of every
16-token phase, per wave group: 0 = end of the L segment's own work (fragments waited for), 1 = out of the barrier
(start of the M segment), 2 = MFMAs issued (base build only), 3 = out of the closing barrier.  Printed: medians of
L work (3 of the previous phase -> 0), barrier wait (0 -> 1), M work (1 -> 2), barrier wait (2 -> 3), whole phase."""
import ctypes as C
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from passt_amd import _lib, ops  # noqa: E402
from passt_amd._lib import PA_BF16  # noqa: E402

SLOTS = 512


def main():
    lib = _lib.load()
    lib.pa_probe_set_buffer.restype = C.c_int
    lib.pa_probe_set_buffer.argtypes = [C.c_void_p]
    buf = torch.zeros(8 * 2 * SLOTS, device="cuda", dtype=torch.int64)
    M, D = 64 * 474, 768
    bf = torch.bfloat16
    rnd = lambda *s: (torch.rand(*s, device="cuda") * 2 - 1).to(bf)   # noqa: E731
    probs = [(rnd(M, 3 * D), rnd(M, D), torch.zeros(3 * D, D, device="cuda"), False),
             (rnd(M, D), rnd(M, D), torch.zeros(D, D, device="cuda"), False),
             (rnd(M, 4 * D), rnd(M, D), torch.zeros(4 * D, D, device="cuda"), False),
             (rnd(M, D), rnd(M, 4 * D), torch.zeros(D, 4 * D, device="cuda"), False)]
    ws = ops.wgrad_tn_batched(probs, PA_BF16)
    torch.cuda.synchronize()
    buf.zero_()
    lib.pa_probe_set_buffer(buf.data_ptr())
    ops.wgrad_tn_batched(probs, PA_BF16, ws)
    torch.cuda.synchronize()
    lib.pa_probe_set_buffer(None)
    st = buf.cpu().numpy().reshape(8, 2, SLOTS)
    out = {}
    for grp in range(2):
        Lw, w1, Mw, w2, ph = [], [], [], [], []
        for wg in range(8):
            s = st[wg, grp][:8 * 3 * 4].reshape(-1, 4).astype(np.int64)
            for k in range(1, len(s)):
                if s[k, 0] == 0 or s[k - 1, 3] == 0:
                    continue
                Lw.append(s[k, 0] - s[k - 1, 3])
                w1.append(s[k, 1] - s[k, 0])
                if s[k, 2]:
                    Mw.append(s[k, 2] - s[k, 1])
                    w2.append(s[k, 3] - s[k, 2])
                else:
                    Mw.append(s[k, 3] - s[k, 1])
                ph.append(s[k, 3] - s[k - 1, 3])
        med = lambda x: float(np.median(x)) if len(x) else None   # noqa: E731
        out[f"group{grp}"] = {"L_work": med(Lw), "wait_after_L": med(w1), "M_work(+wait if no stamp 2)": med(Mw),
                              "wait_after_M": med(w2), "phase": med(ph), "n": len(ph)}
    out["unit"] = "s_memtime ticks (shader cycles, as in profiles/r02_epilogue_probe.json)"
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
