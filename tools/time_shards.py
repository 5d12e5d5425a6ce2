#!/usr/bin/env python3
"""Exact-replay generation shared out over the shards of a multi context (csrc/ldpc_multi.hpp: multi_mt_round), timed on ONE GPU.
All n logical shards sit on the same device, so their kernels share it: if every shard generates the whole tape (LDPC_HIP_MT_SHARDED=0,
round 2's behaviour) the time grows ~n-fold; with the tape shared out the device does the work once, whatever n -- i.e. per-shard
generation work is ~1/n, which on n real GPUs is the wall time.
Usage: python tools/time_shards.py [frames]"""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def worker(mode, frames):
    import numpy as np
    import torch
    import ldpc_lib_amd
    from ldpc_testlib import MS_DEC, load_base_matrix, relift
    from test_mt_replay import seeded_state
    H = relift(load_base_matrix(), 64)
    key, pos = seeded_state(1)
    out = {}
    for n in (1, 2, 4, 8):
        with ldpc_lib_amd.LdpcHipMulti(MS_DEC, H, 64, [0] * n) as m:
            m.mt_set_state(key, pos)
            m.mt_advance(2.0, frames)            # warm-up: buffers, polynomials
            m.mt_frames(2.0, 50, frames)
            torch.cuda.synchronize()
            t = time.perf_counter()
            for _ in range(3):
                m.mt_advance(2.0, frames)
            gen = (time.perf_counter() - t) / 3
            t = time.perf_counter()
            for _ in range(3):
                info, its = m.mt_frames(2.0, 50, frames)
            full = (time.perf_counter() - t) / 3
            out[str(n)] = {"generate_only_ms": gen * 1e3, "noise_decode_count_ms": full * 1e3, "frames_per_s": frames / full,
                           "sharded_rounds_fallbacks": list(m.mt_stats()), "errored": int((info != 0).sum())}
    print("RESULT " + json.dumps(out))


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--worker":
        worker(sys.argv[2], int(sys.argv[3]))
    else:
        frames = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
        res = {"frames_per_call": frames, "what": __doc__.split("Usage")[0].strip()}
        for mode, env in (("tape_shared_out", "1"), ("whole_tape_on_every_shard", "0")):
            e = dict(os.environ, LDPC_HIP_MT_SHARDED=env)
            o = subprocess.run([sys.executable, os.path.abspath(__file__), "--worker", mode, str(frames)], env=e, capture_output=True, text=True, timeout=1500)
            line = [ln for ln in o.stdout.splitlines() if ln.startswith("RESULT ")]
            res[mode] = json.loads(line[0][7:]) if line else {"error": o.stderr[-800:]}
        print(json.dumps(res, indent=1))
