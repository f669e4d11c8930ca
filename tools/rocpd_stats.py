#!/usr/bin/env python
"""Per-kernel summary of a rocprofv3 run (rocpd sqlite database), small enough to commit under profiles/.

    python tools/rocpd_stats.py <results.db> [--steps N] [--skip K] [--sequence N] > stats.txt

Kernel trace: calls, total ms (per step when --steps is given), average / min / max us, share.
PMC runs (rocprofv3 --pmc ...): per kernel the mean of every collected counter per dispatch.
--skip K drops the first K dispatches of every kernel (warm-up / first-touch)."""
import argparse
import re
import sqlite3


def short(name):
    name = re.sub(r"\(.*", "", name)
    name = name.replace("void ", "").replace("pa::", "")
    return name[:70]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("db")
    ap.add_argument("--steps", type=int, default=0)
    ap.add_argument("--skip", type=int, default=0)
    ap.add_argument("--top", type=int, default=40)
    ap.add_argument("--sequence", type=int, default=0, help="also list the last N dispatches in launch order")
    ap.add_argument("--gaps", type=int, default=0, help="GPU idle time: span / busy / idle of the trace and the N largest gaps between "
                                                        "consecutive dispatches with the kernels either side")
    args = ap.parse_args()
    con = sqlite3.connect(args.db)
    rows = con.execute("select s.kernel_name, d.start, d.end, d.dispatch_id, d.grid_size_x, d.workgroup_size_x "
                       "from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on s.id = d.kernel_id "
                       "order by d.start").fetchall()
    per = {}
    for name, st, en, did, gx, wx in rows:
        per.setdefault(name, []).append((en - st, did))
    tot_all = 0
    stats = []
    for name, lst in per.items():
        lst = lst[args.skip:] if len(lst) > args.skip else lst
        ds = [d for d, _ in lst]
        tot = sum(ds)
        tot_all += tot
        stats.append((tot, name, len(ds), tot / len(ds), min(ds), max(ds)))
    stats.sort(reverse=True)
    div = args.steps if args.steps else 1
    print(f"# {args.db}: {len(rows)} dispatches, kernel time total {tot_all / 1e6:.3f} ms" +
          (f" = {tot_all / 1e6 / div:.3f} ms/step over {div} steps" if args.steps else ""))
    print(f"{'kernel':72s} {'calls':>6s} {'ms' + ('/step' if args.steps else ''):>9s} {'avg us':>9s} {'min us':>9s} {'max us':>9s} {'%':>6s}")
    for tot, name, n, avg, mn, mx in stats[:args.top]:
        print(f"{short(name):72s} {n:6d} {tot / 1e6 / div:9.3f} {avg / 1e3:9.2f} {mn / 1e3:9.2f} {mx / 1e3:9.2f} {100.0 * tot / tot_all:6.2f}")
    if args.sequence:
        print(f"\n# last {args.sequence} dispatches in launch order: start offset us, duration us, grid, kernel")
        t0 = rows[-args.sequence][1] if len(rows) >= args.sequence else rows[0][1]
        for name, st, en, did, gx, wx in rows[-args.sequence:]:
            print(f"{(st - t0) / 1e3:10.1f} {(en - st) / 1e3:8.2f} {gx:9d} {short(name)}")
    if args.gaps and rows:
        busy_end, idle, gaps = rows[0][2], 0, []
        for (pn, pst, pen, *_), (name, st, en, *_) in zip(rows, rows[1:]):
            if st > busy_end:
                idle += st - busy_end
                gaps.append((st - busy_end, pn, name, st - rows[0][1]))
            busy_end = max(busy_end, en)
        span = busy_end - rows[0][1]
        print(f"\n# GPU timeline: span {span / 1e6:.3f} ms, idle between dispatches {idle / 1e6:.3f} ms ({100.0 * idle / span:.1f} %)" +
              (f" = {idle / 1e6 / div:.3f} ms/step" if args.steps else "") + f"; gaps > 20 us: {sum(1 for g in gaps if g[0] > 20000)}")
        print(f"# {args.gaps} largest gaps: us idle, at ms, after kernel -> before kernel")
        for g, pn, name, at in sorted(gaps, reverse=True)[:args.gaps]:
            print(f"{g / 1e3:10.1f} {at / 1e6:9.2f}  {short(pn)[:48]:48s} -> {short(name)[:48]}")
    # counters
    try:
        pm = con.execute("select s.kernel_name, p.name, count(*), sum(e.value) from rocpd_pmc_event e "
                         "join rocpd_info_pmc p on p.id = e.pmc_id "
                         "join rocpd_kernel_dispatch d on d.event_id = e.event_id "
                         "join rocpd_info_kernel_symbol s on s.id = d.kernel_id group by s.kernel_name, p.name").fetchall()
    except Exception as e:  # noqa: BLE001
        pm = []
        print("# no counters:", e)
    if pm:
        agg = {}
        for name, cn, n, v in pm:
            agg.setdefault(name, {})[cn] = (n, v)
        print("\n# counters: mean per dispatch")
        order = [s[1] for s in stats]
        for name in order[:args.top]:
            if name in agg:
                print(f"{short(name):72s} " + "  ".join(f"{cn}={v / n:.5g} (n={n})" for cn, (n, v) in sorted(agg[name].items())))


if __name__ == "__main__":
    main()
