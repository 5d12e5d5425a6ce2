#!/usr/bin/env python3
"""Randomised soak of the exact-replay generator: random seeds, start positions inside the block and call splits (single words up
to tens of streams), device samples against the host's std::mt19937 + std::normal_distribution (the oracle's C++), bit for bit.
    python tools/soak_mt.py [cases] [seed]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import ldpc_lib_amd as L  # noqa: E402
from ldpc_testlib import MS_DEC, load_base_matrix, relift  # noqa: E402
from test_mt_replay import oracle_gaussians, seeded_state  # noqa: E402

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
H = relift(load_base_matrix(), 1)
bad = 0
total = 0
with L.LdpcHip(MS_DEC, H, 1) as dec:
    for case in range(cases):
        seed = int(rng.integers(1, 2**31))
        burn = int(rng.integers(0, 5000))
        counts = [int(c) for c in rng.choice([1, 2, 63, 64, 65, 1000, 4097, 52_000, 300_000, 1_500_000, 6_000_000], size=int(rng.integers(1, 7)))]
        want = oracle_gaussians(seed, sum(counts), burn)
        key, pos = seeded_state(seed)
        if burn:
            bg = np.random.MT19937()
            s = bg.state
            s["state"]["key"] = key
            s["state"]["pos"] = pos
            bg.state = s
            bg.random_raw(burn)
            key, pos = bg.state["state"]["key"].astype(np.uint32), int(bg.state["state"]["pos"])
        dec.mt_set_state(key, pos)
        got = torch.cat([dec.mt_normal(c) for c in counts]).cpu().numpy()
        ok = np.array_equal(got.view(np.uint64), want.view(np.uint64))
        bad += not ok
        total += sum(counts)
        print(f"case {case:3d} seed {seed:10d} start {pos:3d} splits {counts} {'ok' if ok else 'MISMATCH'}", flush=True)
print(f"cases {cases}  samples {total}  mismatches {bad}")
sys.exit(1 if bad else 0)
