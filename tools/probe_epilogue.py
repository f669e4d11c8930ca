#!/usr/bin/env python
"""Where does an item of the persistent role-split GEMM spend its time?  (run with the probe library:
PASST_AMD_LIB=passt_amd/libpasst_amd_probe.so python tools/probe_epilogue.py)

For each layer GEMM of the passt_s step (M = 30336) it prints
  * the launch time with the full epilogue / without global traffic in the epilogue / without any epilogue
    (pa_gemm_args.reserved bits 0 / 1, probe build only), and
  * the s_memtime timeline of the items of workgroups 0..7 (per wave group): K-tile durations (the first K-tile
    after an epilogue contains the wait for that epilogue's stores: vmcnt retires in order) and the epilogue span.
"""
import ctypes as C
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from passt_amd import _lib, ops  # noqa: E402
from passt_amd._lib import EPI_DGELU, EPI_GELU, EPI_RESID, EPI_STORE, PA_BF16  # noqa: E402
from bench_kernels import timeit  # noqa: E402

SLOTS = 512
bf = torch.bfloat16


def rnd(*s, dtype=bf):
    return (torch.rand(*s, device="cuda") * 2 - 1).to(dtype)


def main():
    lib = _lib.load()
    lib.pa_probe_set_buffer.restype = C.c_int
    lib.pa_probe_set_buffer.argtypes = [C.c_void_p]
    buf = torch.zeros(8 * 2 * SLOTS, device="cuda", dtype=torch.int64)
    M, D = 64 * 474, 768
    x, h, xf = rnd(M, D), rnd(M, 4 * D), rnd(M, D, dtype=torch.float32)
    out = []
    tunes = [int(v) for v in os.environ.get("PROBE_TUNES", "0").split(",")]
    for name, A, Nn, K, epi in (("qkv store", x, 3 * D, D, EPI_STORE), ("proj resid", x, D, D, EPI_RESID),
                                ("fc1 gelu", x, 4 * D, D, EPI_GELU), ("fc2 resid", h, D, 4 * D, EPI_RESID),
                                ("dfc2 dgelu", x, 4 * D, D, EPI_DGELU), ("dfc1 store", h, D, 4 * D, EPI_STORE)):
        W = rnd(Nn, K) * 0.05
        bias = torch.zeros(Nn, device="cuda")
        if epi == EPI_STORE:
            kw = dict(bias=bias, out_lp=torch.empty(M, Nn, device="cuda", dtype=bf))
        elif epi == EPI_GELU:
            kw = dict(bias=bias, out_lp=torch.empty(M, Nn, device="cuda", dtype=bf),
                      out_lp2=torch.empty(M, Nn, device="cuda", dtype=bf))
        elif epi == EPI_RESID:
            kw = dict(bias=bias, resid=xf, out_f32=torch.empty(M, Nn, device="cuda"))
        else:
            kw = dict(aux=rnd(M, Nn), out_lp=torch.empty(M, Nn, device="cuda", dtype=bf))
        for tune in tunes:
            ops.GEMM_TUNE = tune
            rec = {"gemm": name, "N": Nn, "K": K, "tune": tune}
            lib.pa_probe_set_buffer(None)
            for flag, label in ((0, "full"), (1, "no_stores"), (4, "no_act_math"), (5, "no_stores_no_math"), (2, "no_epilogue")):
                ops.GEMM_RESERVED = flag
                sec = timeit(lambda: ops.gemm_nt(A, W, PA_BF16, epi, **kw), 15)
                rec[label + "_us"] = round(sec * 1e6, 1)
            ops.GEMM_RESERVED = 0
            ops.GEMM_RESERVED = int(os.environ.get("PROBE_STAMP_FLAGS", "0"))
            buf.zero_()
            lib.pa_probe_set_buffer(buf.data_ptr())
            ops.gemm_nt(A, W, PA_BF16, epi, **kw)
            ops.GEMM_RESERVED = 0
            torch.cuda.synchronize()
            lib.pa_probe_set_buffer(None)
            st = buf.cpu().numpy().reshape(8, 2, SLOTS)
            # per item: slot 0 = K loop start, 1+t = end of K-tile t (t < 13), 15 = end of the epilogue
            rows = []
            for wg in range(8):
                for grp in range(2):
                    s = st[wg, grp]
                    for r in range(24):
                        b = s[r * 16:(r + 1) * 16]
                        if b[0] == 0 or b[15] == 0:
                            continue
                        nt = int(np.count_nonzero(b[1:14]))
                        tiles = np.diff(np.concatenate([[b[0]], b[1:1 + nt]])).astype(np.int64)
                        rows.append(dict(wg=wg, grp=grp, round=r, first=int(tiles[0]), second=int(tiles[1]) if nt > 1 else 0,
                                         mid=float(np.median(tiles[2:])) if nt > 3 else 0.0, ntiles=nt,
                                         kloop_stamped=int(b[nt] - b[0]), epi=int(b[15] - b[nt]),
                                         seam=int(s[(r + 1) * 16] - b[15]) if r + 1 < 24 and s[(r + 1) * 16] else 0))
            if rows:
                r0 = [q for q in rows if q["round"] == 0]
                rn = [q for q in rows if q["round"] > 0]
                med = lambda xs, k: float(np.median([q[k] for q in xs])) if xs else 0.0   # noqa: E731
                rec["ticks"] = {"round0": {k: med(r0, k) for k in ("first", "second", "mid", "epi", "seam")},
                                "later": {k: med(rn, k) for k in ("first", "second", "mid", "epi", "seam")},
                                "items": len(rows), "unit": "s_memtime ticks (100 MHz constant clock on gfx9: 10 ns)"}
            out.append(rec)
            print(json.dumps(rec), flush=True)
    ops.GEMM_TUNE = 0
    os.makedirs("gpurun_out", exist_ok=True)
    with open("gpurun_out/probe_epilogue.json", "w") as f:
        json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
