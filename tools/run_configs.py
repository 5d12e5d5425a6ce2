#!/usr/bin/env python3
"""Measure the five BASELINE.json configurations on one MI355X (the 8-GPU ones on the per-GPU shard they would get)
and write profiles/r01_configs.json.  Parity for each configuration is covered by tests/test_gpu_parity.py; this
script only times them (device-side noise, decode + error accounting, inputs resident in HBM)."""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import ldpc_lib_amd as L  # noqa: E402
from ldpc_testlib import load_base_matrix  # noqa: E402

H0 = load_base_matrix()


def measure(name, dec_id, M, snr, B, maxiter, modulation=0, reps=5):
    H = L.relift_base_matrix(H0, M)
    with L.LdpcHip(dec_id, H, M) as dec:
        llr = dec.awgn_llr(snr, 1, 0, B, modulation=modulation)
        cnt = torch.zeros(5, dtype=torch.int64, device="cuda")
        ts = []
        for r in range(reps + 1):
            cnt.zero_()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            hard, iters, _ = dec.decode(llr, maxiter)
            dec.count_errors(hard, iters, counters=cnt)
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
        t = float(np.median(ts[1:]))
        c = cnt.cpu().tolist()
        return {"config": name, "decoder_id": dec_id, "N": dec.N, "K": dec.N - dec.R, "M": M, "max_iterations": maxiter,
                "ebn0_db": snr, "modulation": ["BPSK", "QAM4", "QAM16"][modulation], "frames": B, "kernel": dec.kernel_name,
                "ms": t * 1e3, "frames_per_s": B / t, "frame_iters_per_s": c[4] / t, "fer": c[1] / c[3],
                "ber": c[0] / c[3] / (dec.N - dec.R), "mean_abs_iters": c[4] / c[3]}


out = [
    measure("cfg1 (32,16) min-sum 20 it (plumbing size)", L.DEC_MS, 1, 4.0, 1 << 20, 20),
    measure("cfg2 (2048,1024) min-sum 50 it, operating point", L.DEC_MS, 64, 2.0, 65536, 50),
    measure("cfg2 (2048,1024) min-sum 50 it, worst case (no early exit)", L.DEC_MS, 64, 0.0, 65536, 50),
    measure("cfg3 (2048,1024) sum-product 50 it", L.DEC_SP, 64, 2.0, 32768, 50),
    measure("cfg3 (2048,1024) sum-product 50 it, worst case", L.DEC_SP, 64, 0.0, 8192, 50),
    measure("cfg4 (16384,8192) layered min-sum 50 it (one GPU's shard)", L.DEC_LMS, 512, 1.6, 8192, 50),
    measure("cfg4 (16384,8192) layered min-sum 50 it, worst case", L.DEC_LMS, 512, 0.0, 4096, 50),
    measure("cfg5 (2048,1024) min-sum 50 it behind the 16-QAM soft demapper, Eb/N0 5 dB", L.DEC_MS, 64, 5.0, 65536, 50, modulation=2),
    measure("cfg5 (2048,1024) min-sum 50 it behind the 16-QAM soft demapper, Eb/N0 6 dB", L.DEC_MS, 64, 6.0, 65536, 50, modulation=2),
    measure("f1 (2048,1024) integer min-sum 50 it", L.DEC_IMS, 64, 2.0, 65536, 50),
    measure("(2048,1024) layered min-sum 50 it", L.DEC_LMS, 64, 1.6, 65536, 50),
    measure("f2 shipped search scenario: (4032,2016) M=126 TDMP sum-product (decoder 7) 15 it, 1.7 dB", L.DEC_TASP, 126, 1.7, 32768, 15),
    measure("f2 (2048,1024) TDMP sum-product 15 it", L.DEC_TASP, 64, 1.7, 65536, 15),
    measure("f2 (2048,1024) probability-domain flooding sum-product (decoder 2) 50 it", L.DEC_ASP, 64, 2.0, 32768, 50),
    measure("f2 (2048,1024) Gallager BP, log domain (decoder 0) 50 it, frames chained", L.DEC_BP, 64, 2.0, 32768, 50),
]
# front-end kernels alone
with L.LdpcHip(L.DEC_MS, L.relift_base_matrix(H0, 64), 64) as dec:
    for mod, nm in ((0, "awgn_llr_kernel BPSK"), (2, "awgn_qam_llr_kernel<2> 16-QAM"), (3, "awgn_qam_llr_kernel<3> 64-QAM"), (4, "awgn_qam_llr_kernel<4> 256-QAM")):
        buf = dec.awgn_llr(2.0, 1, 0, 65536, modulation=mod)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            dec.awgn_llr(2.0, 1, 0, 65536, modulation=mod, out=buf)
        torch.cuda.synchronize()
        t = (time.perf_counter() - t0) / 5
        out.append({"config": nm + " 65536 x 2048 LLR (fp64 out)", "ms": t * 1e3, "GB_per_s_written": 65536 * 2048 * 8 / t / 1e9})
for o in out:
    print(json.dumps(o))
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "r01_configs.json"), "w"), indent=1)
