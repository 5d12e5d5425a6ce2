#!/usr/bin/env python3
"""Randomised soak: random protographs x liftings x decoders on the GPU against the CPU oracle (bit-exact decoders only by
default).  Not part of the test-suite (every case compiles its own hiprtc instance); run by hand:
    python tools/soak.py [cases] [seed]"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import ldpc_lib_amd as L  # noqa: E402
from ldpc_testlib import ASP_DEC, BP_DEC, IMS_DEC, LMS_DEC, MS_DEC, SP_DEC, TASP_DEC, Oracle, awgn_llr, pack_bits, random_qc_code  # noqa: E402

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 20
rng = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
FAMILY = (SP_DEC, ASP_DEC, TASP_DEC, BP_DEC) if "--sp" in sys.argv else (MS_DEC, LMS_DEC, IMS_DEC)
GLOBAL = "--global" in sys.argv     # the shape-unlimited tier (ldpc_global.hpp) forced on: no hiprtc, larger shapes too
if GLOBAL:
    os.environ["LDPC_HIP_FORCE_GLOBAL"] = "1"
    FAMILY = (MS_DEC, LMS_DEC, SP_DEC, TASP_DEC, IMS_DEC, ASP_DEC, BP_DEC)
if "--only" in sys.argv:
    FAMILY = (int(sys.argv[sys.argv.index("--only") + 1]),)
TOL = {}   # every decoder bit for bit: exp() / log() are glibc's algorithms on the device
t0 = time.time()
bad = 0
for case in range(cases):
    rh = int(rng.randint(2, 40 if GLOBAL else 13))
    nh = rh + int(rng.randint(2, 40 if GLOBAL else 14))
    M = int(rng.choice([1, 2, 3, 5, 8, 13, 16, 21, 27, 32, 33, 40, 47, 48, 63, 64, 65, 67, 96, 100, 126, 128, 129, 160, 200, 256] + ([300, 513, 700] if GLOBAL else [])))
    weights = tuple(int(x) for x in rng.randint(2, min(rh, 6) + 1, size=4))
    H = random_qc_code(rng, rh, nh, M, weights)
    if (H >= 0).sum(axis=1).max() > 8 and not GLOBAL:
        continue
    frames = 8 if M * nh > 20000 else 24 if M * nh > 2000 else 70
    snrs = tuple(float(x) for x in os.environ["SOAK_SNRS"].split(",")) if os.environ.get("SOAK_SNRS") else (2.0, 5.0)
    llr = np.concatenate([awgn_llr(H, M, s, 500 + case, max(1, frames // len(snrs))) for s in snrs])
    if "--sp" not in sys.argv or "--extremes" in sys.argv:
        llr[0, :3] = [0.0, -0.0, 40000.0]
    for dec_id in FAMILY:
        o = Oracle(H, M)
        d_ref, it_ref, _ = o.decode(dec_id, llr, 30, 0)
        s_ref, _, _ = Oracle(H, M).decode(dec_id, llr, 30, 1)
        try:
            dec = L.LdpcHip(dec_id, H, M)
        except L.LdpcHipError as e:
            print(f"case {case:3d} rh={rh:2d} nh={nh:2d} M={M:3d} dec={dec_id} refused: {str(e)[:90]}", flush=True)
            continue
        with dec:
            hard, iters, soft = dec.decode(torch.from_numpy(llr).cuda(), 30, want_soft=True)
            torch.cuda.synchronize()
            ok = np.array_equal(iters.cpu().numpy(), it_ref) and np.array_equal(hard.cpu().numpy().view(np.uint32), pack_bits(d_ref))
            if dec_id in TOL:
                ok = ok and np.allclose(soft.cpu().numpy(), s_ref, rtol=TOL[dec_id][0], atol=TOL[dec_id][1], equal_nan=True)
            else:
                ok = ok and np.array_equal(soft.cpu().numpy(), s_ref)
            detail = ""
            if not ok:
                it = iters.cpu().numpy()
                hd = hard.cpu().numpy().view(np.uint32)
                sf = soft.cpu().numpy()
                with np.errstate(all="ignore"):
                    rel = np.nanmax(np.abs(sf - s_ref) / np.maximum(np.abs(s_ref), 1e-300))
                detail = (f" iters_equal={np.array_equal(it, it_ref)} hard_equal={np.array_equal(hd, pack_bits(d_ref))} max_rel_soft={rel:.3e}"
                          f" frames_iters_differ={np.flatnonzero(it != it_ref).tolist()[:6]} nan_gpu={int(np.isnan(sf).sum())} nan_ref={int(np.isnan(s_ref).sum())}")
            print(f"case {case:3d} rh={rh:2d} nh={nh:2d} M={M:3d} dec={dec_id} {dec.kernel_name:40s} {'ok' if ok else 'MISMATCH'}{detail}  ({time.time() - t0:.0f}s)", flush=True)
            bad += not ok
print("mismatches:", bad)
sys.exit(1 if bad else 0)
