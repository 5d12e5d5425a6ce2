#!/usr/bin/env python3
"""A/B timing of the min-sum decode variants in ONE process (interleaved rounds, guide rule 24).
usage: python tools/ab_ms.py [variants...]   (LDPC_HIP_MS_VARIANT values: 0 atomics, 1 read-add-write, -1 generic kernel)"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import ldpc_lib_amd  # noqa: E402
from ldpc_testlib import load_base_matrix  # noqa: E402

variants = [int(v) for v in sys.argv[1:]] or [0, 1, -1]
B = int(os.environ.get("AB_FRAMES", "65536"))
snr = float(os.environ.get("AB_SNR", "0.0"))
rounds = int(os.environ.get("AB_ROUNDS", "5"))
H = ldpc_lib_amd.relift_base_matrix(load_base_matrix(), 64)
decs = {}
for v in variants:
    os.environ["LDPC_HIP_MS_VARIANT"] = str(v)
    decs[v] = ldpc_lib_amd.LdpcHip(ldpc_lib_amd.DEC_MS, H, 64)
llr = decs[variants[0]].awgn_llr(snr, 1, 0, B)
ref = None
times = {v: [] for v in variants}
for r in range(rounds + 1):
    for v in variants:
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        hard, iters, _ = decs[v].decode(llr, 50)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        if r:
            times[v].append(dt)
        if ref is None:
            ref = (hard.clone(), iters.clone())
        else:
            assert torch.equal(hard, ref[0]) and torch.equal(iters, ref[1]), f"variant {v} differs"
its = float(ref[1].abs().double().mean())
for v in variants:
    t = np.array(times[v])
    print(f"variant {v:2d}: median {np.median(t)*1e3:8.3f} ms  min {t.min()*1e3:8.3f} ms  -> {B/np.median(t)/1e6:6.3f} Mframes/s "
          f"({B*its/np.median(t)/1e6:7.1f} M frame-iter/s, mean iters {its:.2f})")
