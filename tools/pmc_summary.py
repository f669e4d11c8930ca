#!/usr/bin/env python
"""Summarise a rocprofv3 --pmc run (rocpd sqlite) per kernel: sums of each counter / dispatches."""
import re
import sqlite3
import sys

db = sys.argv[1]
con = sqlite3.connect(db)
cur = con.cursor()
cols = [r[1] for r in cur.execute("pragma table_info(counters_collection)")]
q = ("select k.name, c.counter_name, count(*), sum(c.value) from counters_collection c "
     "join kernels k on k.dispatch_id = c.dispatch_id group by k.name, c.counter_name")
try:
    rows = cur.execute(q).fetchall()
except Exception as e:  # schema differs: print it
    print("schema:", cols, e)
    print([r for r in cur.execute("select * from counters_collection limit 3")])
    sys.exit(0)
agg = {}
for name, cn, n, v in rows:
    agg.setdefault(re.sub(r"\(.*", "", name)[:90], {})[cn] = (n, v)
for k, d in sorted(agg.items()):
    n = max(x[0] for x in d.values())
    print(f"{k}  [{n} samples]")
    print("    " + "  ".join(f"{cn}={v / n:.4g}" for cn, (n_, v) in sorted(d.items())))
