#!/usr/bin/env python3
"""Workload for `rocprofv3 --kernel-trace --stats`: a few rounds of the exact-replay generator (2^27 samples each) and of
noise -> decode -> count on 65536-frame batches.  python tools/prof_gen.py [rounds]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import ldpc_lib_amd  # noqa: E402
from ldpc_testlib import MS_DEC, load_base_matrix, relift  # noqa: E402
from test_mt_replay import seeded_state  # noqa: E402

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
H = relift(load_base_matrix(), 64)
with ldpc_lib_amd.LdpcHip(MS_DEC, H, 64) as dec:
    dec.mt_set_state(*seeded_state(1))
    n = 1 << 27
    buf = torch.empty(n, dtype=torch.float64, device="cuda")
    for _ in range(rounds + 1):
        dec.mt_normal(n, out=buf)
    del buf
    for _ in range(rounds):
        dec.mt_frames(2.0, 50, 65536)
    torch.cuda.synchronize()
print("done")
