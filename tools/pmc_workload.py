#!/usr/bin/env python3
"""One decode launch shape per bench configuration, for the PMC passes (tools/prof_pmc2.sh): the worst-case point (0 dB, every
frame runs all iterations) with the frame count bench.py uses, two launches.  usage: pmc_workload.py <config key>"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench  # noqa: E402
import ldpc_lib_amd  # noqa: E402
from ldpc_testlib import load_base_matrix, relift  # noqa: E402

key = sys.argv[1]
if key == "exact_replay_generator":   # two generation rounds of 2^27 samples of upstream's noise stream (csrc/ldpc_mt.hpp)
    H = relift(load_base_matrix(), 64)
    with ldpc_lib_amd.LdpcHip(bench.DEC_MS, H, 64) as dec:
        k, p = bench.mt19937_seeded(1)
        dec.mt_set_state(k, p)
        buf = torch.empty(1 << 27, dtype=torch.float64, device="cuda")
        for _ in range(2):
            dec.mt_normal(1 << 27, out=buf)
        torch.cuda.synchronize()
        print(key, "samples per round", 1 << 27)
    sys.exit(0)
cfgs = {c["key"]: c for c in bench.EXTRA_CONFIGS}
cfgs["cfg2_min_sum"] = dict(dec=bench.DEC_MS, M=64, frames=bench.FRAMES_PER_GPU, maxiter=50, modulation=0)
c = cfgs[key]
H = relift(load_base_matrix(), c["M"])
with ldpc_lib_amd.LdpcHip(c["dec"], H, c["M"]) as dec:
    llr = dec.awgn_llr(bench.WORST_SNR, 1, 0, c["frames"], modulation=c["modulation"])
    for _ in range(2):
        hard, iters, _ = dec.decode(llr, c["maxiter"])
    torch.cuda.synchronize()
    print(key, dec.kernel_name, "frames", c["frames"], "mean |iters|", float(iters.abs().double().mean()))
