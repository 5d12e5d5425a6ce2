"""Throughput of the exact-replay path (upstream's own mt19937 / normal_distribution noise continued on the device):
samples/s of the generator alone, frames/s of noise -> decode -> count, and the C++ harness end to end (device noise vs host noise).
Usage: python tools/time_exact.py [--frames N] [--host-frames N]"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=400000)
    ap.add_argument("--host-frames", type=int, default=8000)
    ap.add_argument("--snr", type=float, default=2.0)
    args = ap.parse_args()
    import torch
    import ldpc_lib_amd
    from ldpc_testlib import MS_DEC, load_base_matrix, relift
    from test_mt_replay import seeded_state, _compat
    H = relift(load_base_matrix(), 64)
    out = {}
    with ldpc_lib_amd.LdpcHip(MS_DEC, H, 64) as dec:
        key, pos = seeded_state(1)
        dec.mt_set_state(key, pos)
        n = 1 << 27
        buf = torch.empty(n, dtype=torch.float64, device="cuda")
        dec.mt_normal(1 << 20, out=buf)   # warm-up: polynomials, buffers
        dec.mt_normal(n, out=buf)
        torch.cuda.synchronize()
        t = time.perf_counter()
        dec.mt_normal(n, out=buf)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t
        out["generator"] = {"samples": n, "seconds": dt, "samples_per_s": n / dt, "frames_per_s_at_N2048": n / dt / 2048}
        del buf
        for B in (4096, 65536):
            dec.mt_frames(args.snr, 50, B)
            t = time.perf_counter()
            reps = 3
            for _ in range(reps):
                info, its = dec.mt_frames(args.snr, 50, B)
            dt = (time.perf_counter() - t) / reps
            out[f"mt_frames_B{B}"] = {"snr_db": args.snr, "seconds": dt, "frames_per_s": B / dt, "fer": float((info != 0).mean()),
                                      "mean_iters": float(np.abs(its).mean())}
        # the same decode on resident LLRs, for the share of the noise
        llr = dec.mt_llr(args.snr, 65536)
        torch.cuda.synchronize()
        dec.decode(llr, 50)
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(3):
            dec.decode(llr, 50)
        torch.cuda.synchronize()
        out["decode_only_B65536"] = {"seconds": (time.perf_counter() - t) / 3, "frames_per_s": 65536 / ((time.perf_counter() - t) / 3)}
        del llr
    lib = _compat(ldpc_lib_amd)
    Hc = np.ascontiguousarray(H, dtype=np.int32)
    for noise, nfr in (("device", args.frames), ("device_long_run", 10 * args.frames), ("host", args.host_frames)):
        os.environ["LDPC_HIP_EXACT_NOISE"] = noise.split("_")[0]
        res = (C.c_double * 7)()
        nxt = C.c_uint()
        t = time.perf_counter()
        rc = lib.ldpc_bp_simulation_exact_perm(16, 32, Hc.ctypes.data, 64, 50, 10**9, nfr, args.snr, 1.0, MS_DEC, 0, 0, 128, 1, 0, 1, 0,
                                               C.addressof(res), C.addressof(nxt))
        dt = time.perf_counter() - t
        assert rc == 0
        out[f"harness_{noise}_noise"] = {"frames": int(res[5]), "seconds": dt, "frames_per_s": res[5] / dt, "fer": res[1], "errored": int(res[3]),
                                         "rng_next": nxt.value}
    # short runs, as the code search issues them (one bp_simulation call per candidate code and SNR): whole-call latency
    from ldpc_testlib import TASP_DEC
    short = {}
    for name, dec_id, Ms, maxit, n_fe, n_exp, snr in (("cfg2_4001_frames", MS_DEC, 64, 50, 10**9, 4000, 2.0),
                                                       ("shipped_search_scenario_tasp_m126_50_errors", TASP_DEC, 126, 15, 50, 10**8, 1.7),
                                                       ("cfg1_m1_2001_frames", MS_DEC, 1, 20, 10**9, 2000, 4.0)):
        Hs = np.ascontiguousarray(relift(load_base_matrix(), Ms), dtype=np.int32)
        row = {}
        for noise in ("device", "host"):
            os.environ["LDPC_HIP_EXACT_NOISE"] = noise
            best = None
            for rep in range(3):
                res = (C.c_double * 7)()
                nxt = C.c_uint()
                t = time.perf_counter()
                rc = lib.ldpc_bp_simulation_exact_perm(16, 32, Hs.ctypes.data, Ms, maxit, n_fe, n_exp, snr, 1.0, dec_id, 0, 0, 128, 1, 0, 1, 0,
                                                       C.addressof(res), C.addressof(nxt))
                dt = time.perf_counter() - t
                assert rc == 0
                best = dt if best is None or dt < best else best
            row[noise] = {"seconds": best, "frames": int(res[5]), "errored": int(res[3]), "rng_next": nxt.value}
        assert row["device"]["rng_next"] == row["host"]["rng_next"] and row["device"]["errored"] == row["host"]["errored"]
        row["speedup"] = row["host"]["seconds"] / row["device"]["seconds"]
        short[name] = row
    out["whole_call_latency_best_of_3"] = short
    # a code search: one bp_simulation call per candidate matrix the library has never seen (main_good_code_search.cpp:320), shipped
    # scenario settings (TDMP sum-product, 15 iterations, stop at 50 errored frames); hiprtc inside ldpc_hip_open vs in the background
    import tempfile
    search = {}
    base = relift(load_base_matrix(), 64)
    for mode in ("async", "sync"):
        os.environ["LDPC_HIP_JIT"] = mode
        os.environ["LDPC_HIP_EXACT_NOISE"] = "device"
        with tempfile.TemporaryDirectory() as td:
            os.environ["LDPC_HIP_CACHE_DIR"] = td
            per = []
            for k in range(6):
                Hk = base.copy()
                Hk[Hk > 0] = (Hk[Hk > 0] * (17 if mode == "async" else 19) + 3 + k) % 64
                Hk = np.ascontiguousarray(Hk, dtype=np.int32)
                res = (C.c_double * 7)()
                nxt = C.c_uint()
                t = time.perf_counter()
                rc = lib.ldpc_bp_simulation_exact_perm(16, 32, Hk.ctypes.data, 64, 15, 50, 10**8, 1.7, 1.0, TASP_DEC, 0, 0, 128, 1, 0, 1, 0,
                                                       C.addressof(res), C.addressof(nxt))
                assert rc == 0
                per.append({"seconds": time.perf_counter() - t, "frames": int(res[5])})
            search[mode] = {"candidates": per, "mean_seconds": float(np.mean([p["seconds"] for p in per]))}
    del os.environ["LDPC_HIP_JIT"], os.environ["LDPC_HIP_CACHE_DIR"]
    out["code_search_like_loop_tasp_m64"] = search
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
