#!/usr/bin/env python
"""Per-launch HBM traffic of the GEMM family from two rocprofv3 --pmc passes of bench.py
(FETCH_SIZE and WRITE_SIZE must be collected separately: MI355X_MICROARCH.md, rocprofv3 PMC slots).
gfx950 correction: FETCH_SIZE reports 1/2 of the bytes of wide coalesced reads -> x2 (validated here on
ln_fwd_kernel: 93.2 MB read -> 45.5e3 KB reported, and mel_frontend_kernel).  Units: KB."""
import hashlib
import json
import os
import sqlite3
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def sha16(rel):
    with open(os.path.join(ROOT, rel), "rb") as f:
        return hashlib.sha256(f.read()).hexdigest()[:16]


def per_kernel(db, counter):
    con = sqlite3.connect(db)
    cur = con.cursor()
    rows = cur.execute("select k.name, k.dispatch_id, sum(c.value) from counters_collection c join kernels k "
                       "on k.dispatch_id = c.dispatch_id where c.counter_name = ? group by k.name, k.dispatch_id",
                       (counter,)).fetchall()
    out = {}
    for name, _, v in rows:
        out.setdefault(name, []).append(v)
    return out


fetch = per_kernel(sys.argv[1], "FETCH_SIZE")
write = per_kernel(sys.argv[2], "WRITE_SIZE")
res = {}
tot_f = tot_w = n = 0
for name in fetch:
    if "gemm_nt" in name or "gemm_tn" in name:
        f, w = fetch[name], write.get(name, [0])
        res[name.split("(")[0][:80]] = {"launches": len(f), "read_MB_per_launch": round(2 * sum(f) / len(f) / 1e3, 2),
                                         "write_MB_per_launch": round(sum(w) / max(1, len(w)) / 1e3, 2)}
        tot_f += 2 * sum(f)
        tot_w += sum(w) * len(f) / max(1, len(w))
        n += len(f)
# bench.py only reports this number while the kernel source it was measured on is unchanged (committed_traffic())
summary = {"kernel_family": "pa::gemm_nt_* + pa::gemm_tn_* (bf16)", "launches": n, "source_sha16": {"gemm.hip": sha16("passt_amd/csrc/gemm.hip")},
           "hbm_bytes_per_launch": round((tot_f + tot_w) * 1e3 / max(1, n)),
           "read_bytes_per_launch": round(tot_f * 1e3 / max(1, n)), "write_bytes_per_launch": round(tot_w * 1e3 / max(1, n)),
           "method": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes over `bench.py --steps 2 --warmup 1`; "
                     "FETCH_SIZE x2 (gfx950 wide-load correction), WRITE_SIZE as reported; KB -> bytes",
           "per_kernel": res}
print(json.dumps(summary, indent=1))
# the front-end kernel, for bench.py's `frontend.traffic` (north_star: rocprof-reported HBM bytes of the front end)
if len(sys.argv) > 3:
    mf = [v for k, v in fetch.items() if "mel_frontend" in k]
    mw = [v for k, v in write.items() if "mel_frontend" in k]
    if mf and mw:
        f, w = mf[0], mw[0]
        rd, wr = 2 * sum(f) / len(f) * 1e3, sum(w) / len(w) * 1e3
        with open(sys.argv[3], "w") as fh:
            json.dump({"kernel": "pa::mel_frontend_kernel", "launches": len(f), "source_sha16": {"mel.hip": sha16("passt_amd/csrc/mel.hip")},
                       "read_bytes_per_launch": round(rd),
                       "write_bytes_per_launch": round(wr), "hbm_bytes_per_launch": round(rd + wr),
                       "algorithmic_bytes_per_launch": 64 * (320000 + 128 * 1000) * 4,
                       "method": summary["method"]}, fh, indent=1)
