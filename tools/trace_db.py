#!/usr/bin/env python3
"""Per-kernel summary (and, with --timeline K, the launches between the K-th and (K+1)-th mt_finish_kernel) of a rocprofv3
--kernel-trace database (rocpd / sqlite, the default output of this image's rocprofv3).  usage: trace_db.py <results.db> [--timeline K]"""
import collections
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
c = db.cursor()
tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if "kernel_dispatch" in t][0]
ks = [t for t in tabs if "kernel_symbol" in t][0]
rows = list(c.execute(f"select s.kernel_name, d.start, d.end, d.grid_size_x, d.workgroup_size_x from {kd} d join {ks} s on d.kernel_id=s.id order by d.start"))
agg = collections.defaultdict(list)
for n, s, e, g, w in rows:
    agg[n.split("(")[0][:70]].append((e - s) / 1e3)
print(f"{'kernel':72s} {'calls':>5s} {'total ms':>9s} {'avg us':>10s} {'max us':>10s}")
for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
    print(f"{k:72s} {len(v):5d} {sum(v) / 1e3:9.3f} {sum(v) / len(v):10.1f} {max(v):10.1f}")
if "--timeline" in sys.argv:
    K = int(sys.argv[sys.argv.index("--timeline") + 1])
    idx = [i for i, r in enumerate(rows) if "mt_finish" in r[0]]
    i0, i1 = idx[K], idx[K + 1]
    t0 = rows[i0][2]
    for n, s, e, g, w in rows[i0 + 1:i1 + 1]:
        print(f"{(s - t0) / 1e3:9.1f} +{(e - s) / 1e3:8.1f} us  workgroups {g // max(w, 1):6d} x {w:4d}  {n[:60]}")
