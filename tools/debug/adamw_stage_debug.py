import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from passt_amd import ops
from passt_amd._lib import PA_BF16, PA_F32
DEV = "cuda"
TD = {PA_F32: torch.float32, PA_BF16: torch.bfloat16}
for dt in (PA_BF16, PA_F32):
    shapes = [(768, 768), (768,), (2304, 768), (70, 130), (64, 64), (3, 5), (527,), (527, 768), (5000,), (768, 256)]
    copies = ["both", None, "both", "both", "t", "both", None, None, None, "straight"]
    offs, off = [], 0
    for sh in shapes:
        offs.append(off); off += int(np.prod(sh))
    total = off
    g0 = torch.Generator().manual_seed(5)
    P = (torch.rand(total, generator=g0) * 2 - 1).to(DEV)
    G = ((torch.rand(total, generator=g0) * 2 - 1) * 1e-2).to(DEV)
    Mo = ((torch.rand(total, generator=g0) * 2 - 1) * 1e-2).to(DEV)
    Vo = (torch.rand(total, generator=g0) * 1e-4).to(DEV)
    hp = dict(lr=3e-3, b1=0.9, b2=0.999, eps=1e-8, wd=1e-2, step=7)
    pa, ma, va = P.clone(), Mo.clone(), Vo.clone()
    ops.adamw(pa, G, ma, va, hp["lr"], hp["b1"], hp["b2"], hp["eps"], hp["wd"], hp["step"])
    pb, mb, vb = P.clone(), Mo.clone(), Vo.clone()
    entries, outs = [], []
    for sh, o, c in zip(shapes, offs, copies):
        rows = sh[0] if c else 1
        cols = int(np.prod(sh)) // rows
        dst = torch.full((rows, cols), 7.0, device=DEV, dtype=TD[dt]) if c in ("both", "straight") else None
        dst_t = torch.full((cols, rows), 7.0, device=DEV, dtype=TD[dt]) if c in ("both", "t") else None
        entries.append((o, rows, cols, dst, dst_t)); outs.append((dst, dst_t))
    table, n, items = ops.make_adamw_stage_table(entries, DEV)
    print("dt", dt, "n", n, "items", items)
    ops.adamw_stage(pb, G, mb, vb, table, n, items, dt, hp["lr"], hp["b1"], hp["b2"], hp["eps"], hp["wd"], hp["step"])
    torch.cuda.synchronize()
    for sh, o, c, (d, t) in zip(shapes, offs, copies, outs):
        nn = int(np.prod(sh))
        sl = slice(o, o + nn)
        dp = float((pb[sl] - pa[sl]).abs().max()); dm = float((mb[sl] - ma[sl]).abs().max()); dv = float((vb[sl] - va[sl]).abs().max())
        upd = float((pa[sl] - P[sl]).abs().max()); updb = float((pb[sl] - P[sl]).abs().max())
        line = f"  {sh} copies={c} off%4={o % 4}: |p diff| {dp:.3g} |m diff| {dm:.3g} |v diff| {dv:.3g}  update ref {upd:.3g} got {updb:.3g}"
        if d is not None:
            w = pa[sl].view(sh[0], -1).to(TD[dt])
            line += f"  dst diff {float((d.float() - w.float()).abs().max()):.3g} (sevens left {int((d == 7).sum())})"
        if t is not None:
            w = pa[sl].view(sh[0], -1).t().contiguous().to(TD[dt])
            line += f"  dst_t diff {float((t.float() - w.float()).abs().max()):.3g} (sevens left {int((t == 7).sum())})"
        print(line)
