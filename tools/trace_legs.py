#!/usr/bin/env python3
"""rocprofv3 --kernel-trace CSV of a bench.py run -> per-leg summary: bench.py times each configuration as warm-up launches followed
by K timed launches of one decode kernel on one batch shape; the tool's own *_kernel_stats.csv averages ALL launches of a kernel
(warm-up, worst case and operating point together), so this splits the dispatch sequence of every decode kernel into legs (a new
leg starts when the grid size changes or when another decode kernel ran in between) and prints the average duration of the last K
launches of each leg next to bench.py's kernel_ms_avg for the same leg.
usage: trace_legs.py <kernel_trace.csv> <bench.json>"""
import csv
import json
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
bench = json.load(open(sys.argv[2]))
DECODE = ("_spec_", "_body", "flood", "layered", "_chunk_", "_global_")
legs, last = [], None
for r in rows:
    k = r["Kernel_Name"]
    if not any(t in k for t in DECODE) or "coef" in k:
        continue
    key = (k, r["Grid_Size_X"])
    dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
    if key != last:
        legs.append([k.split("(")[0], int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"]), []])
        last = key
    legs[-1][2].append(dur)
# bench.py's launch plan per kernel leg: (label, warm-up launches, timed launches, bench.py's kernel_ms_avg)
plan = {0: [("cfg2 headline, worst case", bench["warmup"], bench["steps"], bench["roofline"]["kernel_ms_avg"]),
            ("cfg2 operating point", 1, max(4, bench["steps"] // 2), None), ("cfg2 FER sweep (ldpc_hip_simulate, 9 points)", 0, 9, None)]}
for i, (name, c) in enumerate(bench.get("configs", {}).items(), start=1):
    if "roofline" in c:
        plan[i] = [(name + ", worst case", 1, c["steps"], c["roofline"]["kernel_ms_avg"]),
                   (name + ", operating point", 1, c["steps"], c["operating_point"]["kernel_ms_avg"])]
print("# rocprofv3 --kernel-trace of `python3 bench.py --no-cpu-baseline`: the dispatches of each decode kernel split into bench.py's legs")
print("# (warm-up launches, then the K timed launches); trace avg = mean End-Start of the K timed dispatches; bench = kernel_ms_avg (HIP events)")
print(f"# {'leg':52s} {'kernel':36s} {'K':>3s} {'trace avg ms':>12s} {'bench ms':>9s}")
for i, (k, wg, d) in enumerate(legs):
    pos = 0
    for label, warm, timed, ms in plan.get(i, []):
        seg = d[pos + warm:pos + warm + timed]
        pos += warm + timed
        if seg:
            print(f"{label:54s} {k:36s} {len(seg):3d} {sum(seg) / len(seg):12.3f} {('%9.3f' % ms) if ms else '        -'}")
    if pos != len(d):
        print(f"#   ({len(d) - pos} further launches of {k} not attributed)")
