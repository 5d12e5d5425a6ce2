#!/usr/bin/env python3
"""rocprofv3 --kernel-trace CSV of a bench.py run -> does the trace agree with bench.py's HIP-event kernel time?  The tool's own
*_kernel_stats.csv averages ALL launches of a kernel (warm-up, worst case, operating point, FER sweep), so this picks out the timed
worst-case launches: bench.py runs the headline first (W warm-up + K timed launches of the flagship), and every other configuration as
1 warm-up + K' timed worst-case launches before its operating point -- the first launches of that configuration's kernel in the trace.
(Configurations that reuse the flagship kernel come after the headline's legs and are skipped here.)
usage: trace_legs.py <kernel_trace.csv> <bench.json of the same run>"""
import collections
import csv
import json
import sys

rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
bench = json.load(open(sys.argv[2]))
seq = collections.defaultdict(list)
for r in rows:
    seq[r["Kernel_Name"].split("(")[0]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
print("# rocprofv3 --kernel-trace of `python3 bench.py --no-cpu-baseline`: mean End-Start of the K timed worst-case dispatches of each")
print("# configuration's decode kernel (trace) next to bench.py's kernel_ms_avg of the same run (HIP events on the launch stream)")
print(f"# {'configuration':28s} {'kernel':36s} {'K':>3s} {'trace avg ms':>12s} {'bench ms':>9s} {'diff %':>7s}")
legs = [("cfg2 headline", bench["roofline"]["kernel"].split(" ")[0], bench["warmup"], bench["steps"], bench["roofline"]["kernel_ms_avg"])]
flag = legs[0][1]
for name, c in bench.get("configs", {}).items():
    if "roofline" in c and c["roofline"]["kernel"].split(" ")[0] != flag:
        legs.append((name, c["roofline"]["kernel"].split(" ")[0], 1, c["steps"], c["roofline"]["kernel_ms_avg"]))
for name, kern, warm, k, ms in legs:
    d = seq.get(kern, [])[warm:warm + k]
    if len(d) != k:
        print(f"{name:30s} {kern:36s} launches not found")
        continue
    avg = sum(d) / k
    print(f"{name:30s} {kern:36s} {k:3d} {avg:12.3f} {ms:9.3f} {100 * (avg - ms) / ms:7.2f}")
