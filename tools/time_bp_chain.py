#!/usr/bin/env python3
"""Gallager BP: cost of chaining frames through upstream's uncleared syndrome array (host-driven fix point) vs independent frames."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import ldpc_lib_amd as L
from ldpc_testlib import load_base_matrix
H = L.relift_base_matrix(load_base_matrix(), 64)
for snr in (2.0, 1.0):
    for chain in (True, False):
        with L.LdpcHip(L.DEC_BP, H, 64) as dec:
            dec.set_bp_chain(chain, True)
            llr = dec.awgn_llr(snr, 1, 0, 32768)
            ts = []
            for r in range(4):
                torch.cuda.synchronize(); t0 = time.perf_counter()
                hard, iters, _ = dec.decode(llr, 50)
                torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
            print(f"BP snr {snr} chain {chain}: {np.median(ts[1:])*1e3:.3f} ms, failed frames {(iters < 0).sum().item()}, mean |it| {iters.abs().double().mean().item():.2f}")
