#!/usr/bin/env python3
"""gpurun_out/pmc2/<config>/<group>/**/counter_collection.csv -> gpurun_out/<ROUND_TAG>_pmc.json (per config: counters per dispatch of the
decode kernel, averaged over its dispatches, + launch shape) keyed by the hash of the kernel sources they were measured on."""
import collections
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

DECODE = ("_spec_", "_body", "flood", "layered", "_chunk_")
cfgs = {c["key"]: c for c in bench.EXTRA_CONFIGS}
cfgs["cfg2_min_sum"] = dict(dec=bench.DEC_MS, M=64, frames=bench.FRAMES_PER_GPU, maxiter=50, modulation=0)
out = {"sources_hash": bench.sources_hash(), "note": "rocprofv3 --pmc, one pass per counter group, worst-case point (0 dB: every frame runs all iterations); "
       "values are per dispatch of the decode kernel; FETCH_SIZE / WRITE_SIZE in KiB as reported (FETCH_SIZE reads half the bytes on gfx950, "
       "profiles/r02_fetch_calibration.txt)", "kernels": {}}
for key, c in cfgs.items():
    acc, n, kname = collections.defaultdict(float), collections.Counter(), None
    for f in glob.glob(os.path.join(ROOT, "gpurun_out", "pmc2", key, "*", "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"]
            if not any(t in k for t in DECODE) or "coef" in k:
                continue
            kname = k
            acc[row["Counter_Name"]] += float(row["Counter_Value"])
            n[row["Counter_Name"]] += 1
    if not acc:
        continue
    waves_per_frame = {bench.DEC_SP: 4, bench.DEC_BP: 8, bench.DEC_ASP: 8}.get(c["dec"], (c["M"] + 63) // 64)
    d = {k: acc[k] / n[k] for k in acc}
    d["kernel"] = kname
    d["frames"] = c["frames"]
    d["iterations"] = c["maxiter"]
    d["waves"] = d.get("SQ_WAVES")
    # wave-iterations = (waves a frame occupies) x frames x iterations each runs at the worst-case point.  The flagship launches
    # PERSISTENT waves that pull frames from a queue, so SQ_WAVES (2048) is not the number of frame-waves any more.
    d["wave_iterations"] = max(d.get("SQ_WAVES") or 0, c["frames"] * waves_per_frame) * c["maxiter"]
    out["kernels"][key] = d
# the exact-replay generator: all ldpc_mt:: kernels of one generation round of 2^27 samples, summed (two rounds were run)
acc, n = collections.defaultdict(float), collections.Counter()
per_kernel = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob(os.path.join(ROOT, "gpurun_out", "pmc2", "exact_replay_generator", "*", "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"]
        if "ldpc_mt::" not in k:
            continue
        acc[row["Counter_Name"]] += float(row["Counter_Value"])
        per_kernel[k.split("(")[0].replace("void ", "")][row["Counter_Name"]] += float(row["Counter_Value"])
if acc:
    rounds = 2
    d = {k: v / rounds for k, v in acc.items()}
    d["samples"] = 1 << 27
    d["per_kernel"] = {k: {c: v / rounds for c, v in cs.items()} for k, cs in per_kernel.items()}
    out["kernels"]["exact_replay_generator"] = d
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", os.environ.get("ROUND_TAG", "r03") + "_pmc.json"), "w"), indent=1)
for k, d in out["kernels"].items():
    if k == "exact_replay_generator":
        print(k, "HBM bytes/sample %.1f" % ((2 * d["FETCH_SIZE"] + d["WRITE_SIZE"]) * 1024 / d["samples"]), "VALU insts/sample %.1f" % (d.get("SQ_INSTS_VALU", 0) * 64 / d["samples"]))
        continue
    print(k, d["kernel"][:50], "VALU/wave-iter %.0f" % (d["SQ_INSTS_VALU"] / d["wave_iterations"]), "LDS/wave-iter %.0f" % (d["SQ_INSTS_LDS"] / d["wave_iterations"]),
          "HBM bytes/frame %.0f" % ((2 * d["FETCH_SIZE"] + d["WRITE_SIZE"]) * 1024 / d["frames"]))
