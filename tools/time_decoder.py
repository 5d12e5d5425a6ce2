#!/usr/bin/env python3
"""Time one decoder configuration on the GPU: python tools/time_decoder.py <dec_id> <M> <snr_db> <frames> [maxiter]"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import ldpc_lib_amd  # noqa: E402
from ldpc_testlib import load_base_matrix  # noqa: E402

dec_id, M, snr, B = int(sys.argv[1]), int(sys.argv[2]), float(sys.argv[3]), int(sys.argv[4])
maxiter = int(sys.argv[5]) if len(sys.argv) > 5 else 50
H = ldpc_lib_amd.relift_base_matrix(load_base_matrix(), M)
dec = ldpc_lib_amd.LdpcHip(dec_id, H, M)
llr = dec.awgn_llr(snr, 1, 0, B)
ts = []
for r in range(4):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    hard, iters, _ = dec.decode(llr, maxiter)
    torch.cuda.synchronize()
    ts.append(time.perf_counter() - t0)
t = float(np.median(ts[1:]))
its = float(iters.abs().double().mean())
fer = float((iters < 0).double().mean())
print(f"dec {dec_id} M {M} N {dec.N} snr {snr} frames {B} [{dec.kernel_name}]: {t*1e3:.3f} ms -> {B/t:,.0f} frames/s, "
      f"{B*its/t/1e6:.2f} M frame-iter/s, mean |iters| {its:.2f}, unconverged {fer:.4f}")
