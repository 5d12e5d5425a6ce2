#!/usr/bin/env python3
"""bench.py -- decoded frames/s of the MI355X min-sum path on BASELINE.json's headline configuration.

    python bench.py --gpus N --steps K --warmup W        (N > 1 without a launcher: this process starts the N ranks itself)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Two multi-GPU routes are timed, on the same workload:
  value / ms_per_step   one process per GPU, torch.distributed over RCCL: decode + count + all-reduce of the 5 counters per step,
                        barrier + synchronize on both sides of the K steps, MAX over ranks (the driver's contract);
  abi_multi             north_star's route: ONE C/C++-style host process, ldpc_hip_open_multi(devices 0..N-1) -- a context, a HIP
                        stream and a host thread per GPU inside libldpc_hip.so -- and ldpc_hip_decode_count_multi per step with ONE
                        grouped ncclAllReduce of 5 x uint64 over xGMI.  Runs in a fresh child process after the ranks are gone.

One "step" = one pass of the hot path over one batch already resident in HBM: ldpc_hip_decode_dev (min-sum, alpha 0.8,
max 50 iterations) on 65536 frames/GPU of the (2048,1024) QC-LDPC code (BASELINE configs[1]) + the error-counter kernel
+ (N > 1) one RCCL all-reduce of the five int64 counters.  Frames are sharded by global frame index, no data-path
collective, per-GPU work fixed => "scaling": "weak".

`value` is measured at the WORST-CASE operating point Eb/N0 = 0 dB, where no frame converges and every frame runs
all 50 iterations (nothing is skipped by early termination); the reference's operating point 2.0 dB (FER ~4 %, early
termination active, as upstream runs it) is reported next to it in "operating_point".

roofline (per kernel, also for every entry of "configs"):
  achieved / peak / frac   SURVEY 8(d)'s ALGORITHMIC message-state bytes per frame-iteration x iterations executed / decode
                           kernel time (HIP events on the launch stream) against the 8 TB/s HBM peak.  The decoders keep that
                           state in VGPRs / LDS for all iterations, so this is an EFFECTIVE figure ("effective": true) and can
                           exceed 1 -- it says how much HBM traffic the on-chip layout avoids, not how busy HBM is.
  traffic / hbm_physical   HBM bytes per launch from rocprofv3 PMC passes (FETCH_SIZE doubled per the gfx950 calibration in
                           profiles/r02_fetch_calibration.txt, + WRITE_SIZE) and the physical HBM fraction they give.
  valu_issue / lds         the BINDING roofs: vector instructions issued per second against the full-rate issue peak
                           (one wave64 VALU instruction per 2 cycles per SIMD, 1024 SIMDs, 2.4 GHz) and LDS-array busy cycles.
PMC figures come from the newest profiles/r*_pmc.json (tools/prof_pmc2.sh) measured on exactly these kernel sources (a hash of
csrc/ is recorded with them); otherwise the fields are null.
"""
import argparse
import hashlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

M, MAXITER, ALPHA = 64, 50, 0.8
FRAMES_PER_GPU = 65536
WORST_SNR, OPER_SNR = 0.0, 2.0
HBM_PEAK_GBS = 8000.0    # MI355X_MICROARCH.md: 8 TB/s spec
CLOCK_HZ = 2.4e9         # max clock
SIMDS, CUS = 1024, 256
VALU_PEAK = SIMDS * CLOCK_HZ / 2.0   # wave64 VALU instructions per second at full rate (2 cycles each on a SIMD-32)

DEC_BP, DEC_SP, DEC_ASP, DEC_MS, DEC_IMS, DEC_TASP, DEC_LMS = 0, 1, 2, 3, 4, 7, 8


def algorithmic_bytes_per_iter(formula, E, R, N):
    """SURVEY 8(d) / BASELINE.md section 3: message-state bytes one decoder iteration touches (4-byte LLR width)."""
    if formula == "ms":
        return 4.25 * E + 30 * R + 8.125 * N
    if formula == "lms":
        return 12.25 * E + 20 * R
    if formula == "sp":
        return 16 * E + 8 * R + 8.125 * N
    return None


# The other configurations of BASELINE.json (and the f1 / f2 rows of SURVEY 8) on one GPU: timed in the same run as the headline
# so that every number in DESIGN.md's table is driver-timed.  `frames` = one GPU's batch; worst case = 0 dB (all iterations run).
EXTRA_CONFIGS = [
    dict(key="cfg3_sum_product", dec=DEC_SP, M=64, frames=16384, maxiter=50, oper_snr=2.0, modulation=0, formula="sp",
         what="(2048,1024) sum-product 50 it (BASELINE configs[2])"),
    dict(key="cfg4_layered_m512", dec=DEC_LMS, M=512, frames=16384, maxiter=50, oper_snr=1.6, modulation=0, formula="lms",
         what="(16384,8192) layered min-sum 50 it, one GPU's shard (BASELINE configs[3])"),
    dict(key="cfg5_qam16_min_sum", dec=DEC_MS, M=64, frames=65536, maxiter=50, oper_snr=5.0, modulation=2, formula="ms",
         what="(2048,1024) min-sum 50 it behind the 16-QAM mapper / soft demapper (BASELINE configs[4])"),
    dict(key="f1_integer_min_sum", dec=DEC_IMS, M=64, frames=65536, maxiter=50, oper_snr=2.0, modulation=0, formula="ms",
         what="(2048,1024) integer min-sum 50 it (SURVEY 8 f1)"),
    dict(key="f2_tasp_m126", dec=DEC_TASP, M=126, frames=16384, maxiter=15, oper_snr=1.7, modulation=0, formula=None,
         what="(4032,2016) M=126 TDMP sum-product 15 it, the shipped search scenario (SURVEY 8 f2)"),
    dict(key="f2_bp_m64", dec=DEC_BP, M=64, frames=8192, maxiter=50, oper_snr=2.0, modulation=0, formula=None,
         what="(2048,1024) Gallager BP (log domain) 50 it, frames chained through upstream's uncleared syndrome (SURVEY 8 f2)"),
    dict(key="f2_asp_m64", dec=DEC_ASP, M=64, frames=16384, maxiter=50, oper_snr=2.0, modulation=0, formula=None,
         what="(2048,1024) flooding sum-product in the probability domain 50 it (SURVEY 8 f2)"),
]


def mt19937_seeded(seed):
    """std::mt19937(seed): the 624 words of init_genrand, next-word index 624 (what `os << generator` prints with libstdc++)."""
    st = np.empty(624, dtype=np.uint64)
    st[0] = seed & 0xffffffff
    for i in range(1, 624):
        st[i] = (1812433253 * (int(st[i - 1]) ^ (int(st[i - 1]) >> 30)) + i) & 0xffffffff
    return st.astype(np.uint32), 624


def exact_replay_block(ldpc_lib_amd, H, device, torch, pmc=None):
    """The exact-replay path (include/ldpc/bp_simulation.h's default): upstream's own noise stream -- std::mt19937 seed 1 through a
    fresh std::normal_distribution per sample, commons_portable.cpp:140,174-178 -- continued ON THE DEVICE (csrc/ldpc_mt.hpp), then
    decode and count.  First the headline configuration's own run (4001 frames at 2.0 dB, 170 errored with
    the upstream binary), then throughput on 65536-frame batches."""
    with ldpc_lib_amd.LdpcHip(DEC_MS, H, M, device=device) as dec:
        key, pos = mt19937_seeded(1)
        # random_codeword() draws (nh - rh) * M = 1024 values of next_random_int(0, 2), one generator word each, before the first
        # noise sample (bp_simulation.cpp:512,160-162): walk the state 1024 words on (numpy's MT19937 is the same recurrence)
        bg = np.random.MT19937()
        s0 = bg.state
        s0["state"]["key"] = key
        s0["state"]["pos"] = pos
        bg.state = s0
        bg.random_raw((H.shape[1] - H.shape[0]) * M)
        dec.mt_set_state(bg.state["state"]["key"].astype(np.uint32), int(bg.state["state"]["pos"]))
        info, its = dec.mt_frames(OPER_SNR, MAXITER, 4001)
        errored = int((info != 0).sum())
        dec.mt_frames(OPER_SNR, MAXITER, FRAMES_PER_GPU)   # warm-up at the batch size (buffers)
        steps = 4
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        nerr = 0
        nit = 0
        for _ in range(steps):
            info, its = dec.mt_frames(OPER_SNR, MAXITER, FRAMES_PER_GPU)
            nerr += int((info != 0).sum())
            nit += int(np.abs(its).sum())
        el = time.perf_counter() - t0
        n = 1 << 27
        buf = torch.empty(n, dtype=torch.float64, device=f"cuda:{device}")
        dec.mt_normal(n, out=buf)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        dec.mt_normal(n, out=buf)
        torch.cuda.synchronize()
        gen_s = time.perf_counter() - t1
        del buf
    words = 4.0 / (np.pi / 4.0)             # mt19937 words per accepted polar attempt
    bytes_per_sample = 2 * 4.0 * words + 8  # words written once and read once (the fused count / scan / emit pass); one fp64 sample out
    traffic = None
    p = (pmc or {}).get("exact_replay_generator")
    if p and p.get("samples") == n:
        traffic = (2.0 * p["FETCH_SIZE"] + p["WRITE_SIZE"]) * 1024.0   # per round of n samples, FETCH_SIZE x2 per the gfx950 calibration
    return {
        "what": "upstream's mt19937 + normal_distribution noise stream continued on the device -> min-sum decode -> count "
                "(ldpc_hip_mt_frames), (2048,1024), 50 it, Eb/N0 2.0 dB, seed 1: bit-identical to the frame-by-frame host loop",
        "headline_run": {"frames": 4001, "errored_frames": errored, "upstream_binary": "170 / 4001 (BASELINE.md section 2)"},
        "value": steps * FRAMES_PER_GPU / el, "unit": "frames/s", "frames_per_step": FRAMES_PER_GPU, "steps": steps,
        "fer": nerr / (steps * FRAMES_PER_GPU), "mean_iters_per_frame": nit / (steps * FRAMES_PER_GPU),
        "generator": {"samples_per_s": n / gen_s, "samples": n, "bound": "hbm", "algorithmic_bytes_per_sample": bytes_per_sample,
                      "achieved_GBs": n / gen_s * bytes_per_sample / 1e9, "frac": n / gen_s * bytes_per_sample / 1e9 / HBM_PEAK_GBS,
                      "traffic": traffic, "traffic_unit": "HBM bytes per 2^27-sample round (PMC)",
                      "hbm_physical_frac": traffic / gen_s / 1e9 / HBM_PEAK_GBS if traffic else None},
    }


def sources_hash():
    """Hash of the kernel sources: PMC figures in profiles/r*_pmc.json are only quoted for the code they were measured on."""
    h = hashlib.sha256()
    d = os.path.join(ROOT, "ldpc-lib_amd", "csrc")
    for sub in ("", "aot"):
        for fn in sorted(os.listdir(os.path.join(d, sub))):
            p = os.path.join(d, sub, fn)
            if os.path.isfile(p) and fn.endswith((".hpp", ".hip")):
                h.update(fn.encode())
                h.update(open(p, "rb").read())
    return h.hexdigest()[:16]


def load_pmc():
    """the newest profiles/r*_pmc.json that was measured on exactly these kernel sources"""
    import glob
    h = sources_hash()
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc.json")), reverse=True)
    seen = []
    for f in files:
        try:
            pj = json.load(open(f))
        except Exception:
            continue
        if pj.get("sources_hash") == h:
            kern = pj.get("kernels", {})
            kern["_file"] = os.path.relpath(f, ROOT)
            return kern, None
        seen.append(f"{os.path.basename(f)} ({pj.get('sources_hash')})")
    return {}, f"no profiles/r*_pmc.json was measured on these sources ({h}); have: {', '.join(seen) or 'none'}"


def roofline_block(kernel_name, kernel_ms_avg, launches, sum_iters_per_launch, frames_per_launch, bytes_iter, pmc, pmc_why, pmc_key):
    kern_s = kernel_ms_avg / 1e3
    achieved = (sum_iters_per_launch * bytes_iter) / kern_s / 1e9 if (bytes_iter and kern_s > 0) else None
    r = {
        "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS if achieved else None,
        "effective": True, "traffic": None, "traffic_unit": "bytes per launch", "traffic_source": pmc_why,
        "kernel": kernel_name, "kernel_ms_avg": kernel_ms_avg, "launches": launches,
        "algorithmic_bytes_per_frame_iter": bytes_iter,
        "algorithmic_bytes_per_launch": sum_iters_per_launch * bytes_iter if bytes_iter else None,
        "binding_roof": None, "hbm_physical": None, "valu_issue": None, "lds": None,
    }
    p = pmc.get(pmc_key)
    if p and kern_s > 0 and p.get("frames") == frames_per_launch:
        hbm = (2.0 * p["FETCH_SIZE"] + p["WRITE_SIZE"]) * 1024.0
        wave_iters = p["wave_iterations"]
        r["traffic"] = hbm
        r["traffic_source"] = f"{pmc.get('_file')} (rocprofv3 --pmc, one pass per counter group, same launch shape; FETCH_SIZE x2 per profiles/r02_fetch_calibration.txt)"
        r["hbm_physical"] = {"GBps": hbm / kern_s / 1e9, "frac": hbm / kern_s / 1e9 / HBM_PEAK_GBS, "bytes_per_frame": hbm / frames_per_launch}
        r["valu_issue"] = {"valu_insts_per_launch": p["SQ_INSTS_VALU"], "valu_insts_per_wave_iter": p["SQ_INSTS_VALU"] / wave_iters,
                           "cycles_per_inst": kern_s * CLOCK_HZ * SIMDS / p["SQ_INSTS_VALU"],
                           "achieved_ginst_s": p["SQ_INSTS_VALU"] / kern_s / 1e9, "peak_ginst_s": VALU_PEAK / 1e9,
                           "frac": p["SQ_INSTS_VALU"] / kern_s / VALU_PEAK}
        r["lds"] = {"lds_insts_per_wave_iter": p["SQ_INSTS_LDS"] / wave_iters, "frac": p["SQ_LDS_IDX_ACTIVE"] / (kern_s * CLOCK_HZ * CUS),
                    "bank_conflict_cycles": p["SQ_LDS_BANK_CONFLICT"]}
        r["binding_roof"] = "valu_issue" if r["valu_issue"]["frac"] >= r["lds"]["frac"] else "lds"
    return r


def cpu_baseline(H, seconds_budget=12.0):
    """CPU decode-only timing on this host, on a bounded sample of the SAME workload (0 dB, 50 iterations/frame).
    Uses the compiled upstream reference (oracle/_ref) when it travelled with the repo, else the C restatement.
    Runs BEFORE anything touches the GPU (the workers are forked; a process that has initialised HIP must not fork)."""
    import multiprocessing as mp

    from ldpc_testlib import MS_DEC, Oracle, Reference, awgn_llr, ref_lib

    kind = "reference" if ref_lib() is not None else "port"
    cores = max(1, min(os.cpu_count() or 1, 32))
    probe = awgn_llr(H, M, WORST_SNR, 1, 8)
    dec = Reference(MS_DEC, H, M) if kind == "reference" else Oracle(H, M)
    dec.decode(MS_DEC, probe, MAXITER, 0)   # cold pass (page-in, first-touch): not timed, the sample is sized from the warm one
    t0 = time.perf_counter()
    dec.decode(MS_DEC, probe, MAXITER, 0)
    per_frame = (time.perf_counter() - t0) / 8
    n1 = max(16, int(seconds_budget / 2 / per_frame))
    llr = awgn_llr(H, M, WORST_SNR, 2, n1)
    t0 = time.perf_counter()
    _, its, _ = dec.decode(MS_DEC, llr, MAXITER, 0)
    t1 = time.perf_counter() - t0
    one_core = n1 / t1

    def work(q, n):
        d = Reference(MS_DEC, H, M) if kind == "reference" else Oracle(H, M)
        x = llr[:n]
        t = time.perf_counter()
        d.decode(MS_DEC, x, MAXITER, 0)
        q.put(time.perf_counter() - t)

    all_core = None
    if cores > 1:
        ctx = mp.get_context("fork")
        q = ctx.Queue()
        n_each = max(8, min(n1, int(seconds_budget / 2 / per_frame)))
        ps = [ctx.Process(target=work, args=(q, n_each)) for _ in range(cores)]
        t0 = time.perf_counter()
        for p in ps:
            p.start()
        for p in ps:
            p.join()
        wall = time.perf_counter() - t0
        all_core = cores * n_each / wall
    return {
        "value": all_core if all_core is not None else one_core, "unit": "frames/s",
        "cores": cores if all_core is not None else 1, "kind": kind, "value_1core": one_core,
        "sample": f"{n1} frames single-thread + {cores}x same frames one process per core, (2048,1024) min-sum, "
                  f"Eb/N0 {WORST_SNR} dB, all {MAXITER} iterations run (mean |iters| {float(np.abs(its).mean()):.1f}), decode only, "
                  "measured before the GPU was initialised",
    }


def abi_multi_leg(n, steps, warmup, frames):
    """north_star's multi-GPU route, timed: one host process, ldpc_hip_open_multi over n devices (a context + HIP stream + host
    thread per GPU inside the library), per step ldpc_hip_decode_count_multi on batches resident in HBM = min-sum decode + error
    count on every GPU, then ONE grouped RCCL all-reduce of the five uint64 counters.  The call is synchronous (the counters come
    back to the host), so wall time around K calls is the whole-job time.  With fewer GPUs than shards (rehearsal on a one-GPU box)
    shards share devices and the counters are summed on the host -- `reduction` says which."""
    import torch

    import ldpc_lib_amd
    from ldpc_testlib import load_base_matrix, relift
    H = relift(load_base_matrix(), M)
    ndev = torch.cuda.device_count()
    if ndev < 1:
        raise RuntimeError("no GPU")
    devices = [i % ndev for i in range(n)]
    with ldpc_lib_amd.LdpcHipMulti(DEC_MS, H, M, devices) as m:
        nb = max(1, min(steps, 4))
        while nb > 1 and nb * frames * m.N * 8 * max(1, -(-n // ndev)) > 24e9:
            nb -= 1
        batches = []
        for s in range(nb):   # distinct frames per step and shard: global frame index = (s*n + i)*frames + f
            row = []
            for i, d in enumerate(devices):
                t = torch.empty((frames, m.N), dtype=torch.float64, device=f"cuda:{d}")
                m.channel_llr(i, t, WORST_SNR, 1, (s * n + i) * frames)
                row.append(t)
            batches.append(row)
        for d in set(devices):
            torch.cuda.synchronize(d)
        for s in range(warmup):
            m.decode_count(batches[s % nb], MAXITER, first_frame=(s % nb) * n * frames, alpha=ALPHA)
        tot = {"nse": 0, "nde": 0, "nue": 0, "frames": 0, "sum_abs_iters": 0}
        t0 = time.perf_counter()
        for s in range(steps):
            c = m.decode_count(batches[s % nb], MAXITER, first_frame=(s % nb) * n * frames, alpha=ALPHA)
            for k in tot:
                tot[k] += c[k]
        el = time.perf_counter() - t0
        assert tot["frames"] == frames * n * steps, (tot, frames, n, steps)
        return {
            "what": "one host process, ldpc_hip_open_multi + ldpc_hip_decode_count_multi (csrc/ldpc_multi.hpp): per step min-sum decode "
                    "+ error count of a resident batch on every GPU, one grouped all-reduce of the counters",
            "value": tot["frames"] / el, "unit": "frames/s", "n_gpus": n, "devices": devices, "steps": steps, "warmup": warmup,
            "ms_per_step": el / steps * 1e3, "frames_per_gpu_per_step": frames, "reduction": m.reduction,
            "collective": "rccl ncclAllReduce 5 x uint64, one ncclGroupStart/End per step" if m.reduction == "rccl" else
                          ("host sum of the five counters (shards share a device)" if n > 1 else None),
            "comm_inits": int(m.lib.ldpc_hip_multi_comm_inits()),
            "fer": tot["nde"] / tot["frames"], "mean_iters_per_frame": tot["sum_abs_iters"] / tot["frames"],
        }


def clean_env():
    """this process's environment without a launcher's rendezvous variables: a child that starts its own torch.distributed job must not
    inherit them -- with TORCHELASTIC_USE_AGENT_STORE=True (set by torch.distributed.run) rank 0 would not host the store and every
    rank of the child job would wait for a server that does not exist"""
    drop = ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "GROUP_RANK", "ROLE_RANK", "ROLE_NAME", "LOCAL_WORLD_SIZE",
            "ROLE_WORLD_SIZE", "GROUP_WORLD_SIZE")
    env = {k: v for k, v in os.environ.items() if k not in drop and not k.startswith("TORCHELASTIC_")}
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return env


def run_abi_multi_child(args, timeout=900, leg="abi_multi"):
    """the abi_multi (or exact_ranks) leg in a fresh child process (this one may hold a torch.distributed rank's GPU state, or -- the
    launcher -- must never touch the GPU); returns its JSON or {"error": ...}"""
    import subprocess
    env = clean_env()
    cmd = [sys.executable, os.path.abspath(__file__), "--leg", leg, "--gpus", str(args.gpus), "--steps", str(args.steps),
           "--warmup", str(args.warmup), "--frames", str(args.frames), "--backend", args.backend]
    try:
        p = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True, timeout=timeout)
        lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
        if p.returncode != 0 or not lines:
            return {"error": f"{leg} leg exited with {p.returncode}", "stdout_tail": p.stdout[-500:]}
        return json.loads(lines[-1])
    except Exception as ex:
        return {"error": repr(ex)}


EXACT_RANKS_ROUND = 65536   # global frames per round: the tape of one round is shared out over the ranks


def exact_ranks_worker(args):
    """one rank of the exact_ranks leg: host.bp_simulation(exact_seed=1) in a torch.distributed job, the generator's tape of every
    65536-frame round shared out over the ranks (ldpc_hip_mt_shard_*), every rank decoding its 1/N slice.  Strong scaling by
    construction (the round is the reference loop's, not ours to grow).  Counters and the end state of the generator are printed:
    they must not depend on N."""
    import zlib

    import torch
    import torch.distributed as dist

    import ldpc_lib_amd
    from ldpc_lib_amd import host
    from ldpc_testlib import load_base_matrix, relift
    world, rank, local = int(os.environ["WORLD_SIZE"]), int(os.environ["RANK"]), int(os.environ["LOCAL_RANK"])
    ndev = torch.cuda.device_count()
    if ndev < 1:
        sys.exit(f"bench.py exact_ranks rank {rank}/{world}: no GPU visible")
    backend = args.backend if world <= ndev else "gloo"   # fewer GPUs than ranks: a rehearsal, ranks share devices
    local %= ndev
    torch.cuda.set_device(local)
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group("gloo")
    H = relift(load_base_matrix(), M)
    rounds = max(4, min(args.steps, 16))
    per_rank = EXACT_RANKS_ROUND // world   # N not a power of two: one short extra round, a roll-back inside it
    src = host.MtFrameSource(H, M, DEC_MS, MAXITER, OPER_SNR, 0, 0, 1, local, ALPHA)
    try:
        def run(k):
            return ldpc_lib_amd.bp_simulation(H, M, MAXITER, 1 << 40, k * EXACT_RANKS_ROUND - 1, OPER_SNR, 1.0, decoder_type=DEC_MS,
                                              batch=per_rank, device=local, alpha=ALPHA, return_state=True, source=src)
        run(2)                                                 # warm-up: buffers, jump polynomials, communicator
        src.dec.mt_set_state(*host.mt19937_state(1, (H.shape[1] - H.shape[0]) * M))
        src.dec.mt_set_frame_index(0)
        src.shared_rounds = src.fallback_rounds = 0
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        t0 = time.perf_counter()
        ber, fer, st = run(rounds)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        el = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
        if world > 1:
            if backend == "nccl":
                el = el.cuda()
            dist.all_reduce(el, op=dist.ReduceOp.MAX)
        el = float(el.item())
    finally:
        src.close()
    if world > 1:
        dist.destroy_process_group()
    if rank == 0:
        words, pos = st["generator"]
        print(json.dumps({
            "what": "host.bp_simulation(exact_seed=1), one process per GPU: per 65536-frame round every rank makes ~1/N of the generator's "
                    "tape (ldpc_hip_mt_shard_begin/emit/commit; two all-gathers of a few integers + one 2.5 KB broadcast), decodes its "
                    "1/N of the frames, and the 8-byte records are all-gathered; strong scaling, results independent of N",
            "value": st["experiment"] / el, "unit": "frames/s", "n_gpus": world, "backend": backend if world > 1 else None,
            "rounds": rounds, "frames_per_round": per_rank * world, "ms_per_round": el / rounds * 1e3, "ebn0_db": OPER_SNR,
            "frames": st["experiment"], "errored_frames": st["nde"], "bit_errors": st["nse"], "sum_iterations": st["sum_abs_iters"],
            "generator_state_crc32": zlib.crc32(words.tobytes()) & 0xffffffff, "generator_next_index": int(pos),
            "tape_shared_rounds": st["tape_shared_rounds"], "tape_fallback_rounds": st["tape_fallback_rounds"]}))


def exact_ranks_leg(args, timeout=420):
    """launcher of the exact_ranks leg: a process without GPU state starts the N ranks, waits for them with a deadline and relays
    rank 0's line; on any failure the ranks it started are killed and the error is the result"""
    import socket
    import subprocess
    n = args.gpus
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, os.path.abspath(__file__), "--leg", "exact_ranks", "--spawned", "--gpus", str(n), "--steps", str(args.steps),
           "--backend", args.backend]
    procs = []
    for r in range(n):
        env = dict(clean_env(), RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port))
        procs.append(subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, text=(r == 0)))
    deadline, failed = time.time() + timeout, None
    while failed is None and any(p.poll() is None for p in procs):
        for r, p in enumerate(procs):
            if p.poll() not in (None, 0):
                failed = f"rank {r} exited with {p.returncode}"
        if time.time() > deadline:
            failed = "timeout"
        time.sleep(0.1)
    if failed is not None:
        for p in procs:   # exactly the processes started above
            if p.poll() is None:
                p.kill()
        return {"error": failed}
    lines = [ln for ln in procs[0].stdout.read().splitlines() if ln.startswith("{")]
    return json.loads(lines[-1]) if lines else {"error": "rank 0 printed no result line"}


def launch_ranks(args):
    """`python bench.py --gpus N` with no launcher: this process -- which never touches the GPU -- starts the N ranks as fresh child
    processes (one per GPU, torch.distributed rendezvous on 127.0.0.1), relays rank 0's JSON line with the abi_multi leg (a further
    fresh child, after the ranks are gone) added, and exits non-zero if any rank fails."""
    import socket
    import subprocess
    n = args.gpus
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, os.path.abspath(__file__), "--gpus", str(n), "--steps", str(args.steps), "--warmup", str(args.warmup),
           "--frames", str(args.frames), "--backend", args.backend, "--spawned", "--no-cpu-baseline"]
    if args.no_extras:
        cmd.append("--no-extras")
    procs = []
    for r in range(n):
        env = dict(clean_env(), RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port))
        procs.append(subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, text=(r == 0)))
    deadline = time.time() + float(os.environ.get("LDPC_BENCH_RANK_TIMEOUT", "1500"))
    failed = None
    while failed is None and any(p.poll() is None for p in procs[1:]):
        for r, p in enumerate(procs[1:], 1):
            if p.poll() not in (None, 0):
                failed = f"rank {r} exited with {p.returncode}"
        if time.time() > deadline:
            failed = "timeout"
        if procs[0].poll() not in (None, 0):
            failed = f"rank 0 exited with {procs[0].returncode}"
        time.sleep(0.2)
    out0 = ""
    if failed is None:
        try:
            out0, _ = procs[0].communicate(timeout=max(1.0, deadline - time.time()))
            if procs[0].returncode != 0:
                failed = f"rank 0 exited with {procs[0].returncode}"
        except subprocess.TimeoutExpired:
            failed = "timeout"
    if failed is not None:
        for p in procs:   # exactly the processes started above
            if p.poll() is None:
                p.kill()
        print(f"bench.py: {failed}", file=sys.stderr)
        return 1
    lines = [ln for ln in out0.splitlines() if ln.startswith("{")]
    if not lines:
        print("bench.py: rank 0 printed no result line", file=sys.stderr)
        return 1
    out = json.loads(lines[-1])
    out["launcher"] = "bench.py (parent process without GPU state started one fresh process per rank)"
    out["abi_multi"] = run_abi_multi_child(args)
    if not args.no_extras:
        out["exact_replay_ranks"] = exact_ranks_leg(args)
    print(json.dumps(out))
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--frames", type=int, default=FRAMES_PER_GPU, help="frames per GPU per step")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip operating point, FER sweep and the other configurations (profiling runs)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="nccl (= RCCL over xGMI, the measured configuration); gloo only rehearses the N>1 code path on a box "
                         "with fewer GPUs than ranks (ranks then share devices and the counters are reduced on the host)")
    ap.add_argument("--leg", default=None, choices=["abi_multi", "exact_ranks"],
                    help="internal: run only the C-ABI multi-device leg / the rank-sharded exact-replay leg and print its JSON")
    ap.add_argument("--spawned", action="store_true", help="internal: this rank was started by bench.py itself (the parent adds the abi_multi leg)")
    args = ap.parse_args()

    if args.leg == "abi_multi":
        print(json.dumps(abi_multi_leg(args.gpus, args.steps, args.warmup, args.frames)))
        return
    if args.leg == "exact_ranks":
        if args.spawned:
            exact_ranks_worker(args)
        else:
            print(json.dumps(exact_ranks_leg(args)))
        return
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(launch_ranks(args))

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))

    from ldpc_testlib import load_base_matrix, relift
    H = relift(load_base_matrix(), M)

    # the CPU leg first: nothing has touched the GPU yet (no torch.cuda call, libldpc_hip.so not loaded)
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        try:
            cpu = cpu_baseline(H)
        except Exception as ex:  # the bench line must still be printed
            cpu = {"value": None, "error": repr(ex)}

    import torch
    import torch.distributed as dist

    import ldpc_lib_amd

    if world != args.gpus:
        sys.exit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    ndev = torch.cuda.device_count()
    if ndev < 1:
        sys.exit(f"bench.py rank {rank}/{world}: no GPU visible -- the decoder has no CPU fallback")
    if args.backend == "nccl" and world > ndev:
        sys.exit(f"{world} ranks but {ndev} GPUs: one rank per GPU is required with RCCL")
    local = local % max(ndev, 1)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group("gloo")

    def all_reduce(t, op=None):
        """in-place sum (or `op`) over ranks; RCCL on the device tensor, or via the host for the gloo rehearsal"""
        kw = {} if op is None else {"op": op}
        if args.backend == "nccl":
            dist.all_reduce(t, **kw)
        else:
            h = t.cpu()
            dist.all_reduce(h, **kw)
            t.copy_(h)

    def barrier():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    pmc, pmc_why = load_pmc()

    def timed(dec, B, snr, steps, warmup, maxiter, modulation=0, dec_b=None):
        """W warm-up + K timed steps (decode + count [+ all-reduce]) over batches resident in HBM; returns wall seconds (max over
        ranks), the counter totals and the decode kernel's HIP-event time.  With a second context `dec_b` consecutive steps alternate
        between two HIP streams (one context each), so that two batches are in flight: where frames leave early (operating point) the
        slots the last long-running frames of one batch leave idle are taken by the next batch."""
        nb = min(max(steps, 1), 8)
        if B * dec.N * 8 * nb > 12e9:
            nb = max(1, int(12e9 // (B * dec.N * 8)))
        # distinct frames per step and per rank: global frame index = (step*world + rank)*B + i
        batches = [dec.awgn_llr(snr, seed=1, first_frame=(s * world + rank) * B, B=B, modulation=modulation) for s in range(nb)]
        lanes = []
        for d in ([dec] if dec_b is None else [dec, dec_b]):
            lanes.append({"dec": d, "stream": torch.cuda.current_stream(dev) if dec_b is None else torch.cuda.Stream(dev),
                          "hard": torch.empty((B, dec.hard_words), dtype=torch.int32, device=dev),
                          "iters": torch.empty((B,), dtype=torch.int32, device=dev),
                          "cnt": torch.zeros(5, dtype=torch.int64, device=dev), "tot": torch.zeros(5, dtype=torch.int64, device=dev)})
        torch.cuda.synchronize(dev)

        def step(s):
            ln = lanes[s % len(lanes)]
            with torch.cuda.stream(ln["stream"]):
                ln["dec"].decode(batches[s % nb], maxiter, alpha=ALPHA, out=(ln["hard"], ln["iters"], None))
                ln["cnt"].zero_()
                ln["dec"].count_errors(ln["hard"], ln["iters"], counters=ln["cnt"])
                if world > 1:
                    all_reduce(ln["cnt"])  # 40-byte message: {nse, nde, nue, frames, sum|iters|}
                ln["tot"].add_(ln["cnt"])

        for s in range(warmup):
            step(s)
        barrier()
        for ln in lanes:
            ln["tot"].zero_()
            ln["dec"].profile(True)
            ln["dec"].profile_read(reset=True)
        barrier()
        t0 = time.perf_counter()
        for s in range(steps):
            step(s)
        barrier()
        el = time.perf_counter() - t0
        kms, klaunch = 0.0, 0
        for ln in lanes:
            k, l = ln["dec"].profile_read(reset=True)
            kms, klaunch = kms + k, klaunch + l
            ln["dec"].profile(False)
        t = torch.tensor([el], dtype=torch.float64, device=dev)
        if world > 1:
            all_reduce(t, op=dist.ReduceOp.MAX)
        tot = sum(ln["tot"] for ln in lanes)
        del batches
        return float(t.item()), tot.cpu().tolist(), kms, klaunch

    # ---------------- headline: BASELINE configs[1]
    B = args.frames
    dec = ldpc_lib_amd.LdpcHip(DEC_MS, H, M, device=local)
    N, E, R = dec.N, dec.edges * M, dec.R
    bytes_iter = algorithmic_bytes_per_iter("ms", E, R, N)
    el, tot, kms, klaunch = timed(dec, B, WORST_SNR, args.steps, args.warmup, MAXITER)
    frames_total = tot[3]
    assert frames_total == B * args.steps * world, (frames_total, B, args.steps, world)
    value = frames_total / el
    sum_iters_rank = tot[4] / world  # every rank does the same amount of work (weak scaling)

    out = {
        "metric": "decoded frames/sec, (2048,1024) QC-LDPC, 50 iters min-sum", "value": value, "unit": "frames/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": el / args.steps * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {
            "workload": "files/input32_16.jsonx code (SURVEY Appendix C base matrix 16x32, 112 circulants) lifted to "
                        "(2048,1024) M=64, flooding min-sum alpha=0.8, max 50 iterations, all-zero codeword BPSK/AWGN "
                        f"Eb/N0 {WORST_SNR} dB (every frame runs all 50 iterations), LLR fp64 resident in HBM",
            "frames_per_gpu_per_step": B, "global_frames_per_step": B * world, "sharding": f"frames x{world}", "collective": (args.backend + " all-reduce of 5 int64 counters per step") if world > 1 else None,
            "fer": tot[1] / tot[3], "mean_iters_per_frame": tot[4] / tot[3],
        },
        "roofline": roofline_block(dec.kernel_name, kms / max(klaunch, 1), klaunch, sum_iters_rank / max(klaunch, 1), B, bytes_iter, pmc, pmc_why, "cfg2_min_sum"),
    }
    out["roofline"]["note"] = ("achieved = effective message-state bandwidth (SURVEY 8d): the state is VGPR/LDS resident, compulsory HBM bytes per frame = "
                               f"{8 * N} (fp64 LLR in) + {N // 8 + 4} (packed bits + iters out); the binding roof is VALU issue "
                               "(fp64 + half-rate VOP3 / compare / select), see valu_issue / lds / hbm_physical")

    if not args.no_extras:
        osteps = max(4, args.steps // 2)
        el2, tot2, kms2, kl2 = timed(dec, B, OPER_SNR, osteps, 1, MAXITER)
        with ldpc_lib_amd.LdpcHip(DEC_MS, H, M, device=local) as dec_b:   # the same workload with two batches in flight on two streams
            el3, tot3, kms3, kl3 = timed(dec, B, OPER_SNR, 2 * osteps, 2, MAXITER, dec_b=dec_b)
        worst_fi = sum_iters_rank * world / el   # frame-iterations/s with no early exit
        out["operating_point"] = {
            "ebn0_db": OPER_SNR, "value": tot3[3] / el3, "unit": "frames/s", "fer": tot3[1] / tot3[3],
            "ber": tot3[0] / tot3[3] / (N - R), "mean_iters_per_frame": tot3[4] / tot3[3],
            "how": "two batches in flight on two HIP streams (one context each): the slots that the last long-running frames of a "
                   "batch leave idle are taken by the next batch; persistent waves pull frames from a queue (SpecArgs::queue)",
            "frame_iterations_per_s": tot3[4] / el3, "fraction_of_worst_case_frame_iterations_per_s": (tot3[4] / el3) / worst_fi,
            "one_batch_at_a_time": {"value": tot2[3] / el2, "frame_iterations_per_s": tot2[4] / el2,
                                    "fraction_of_worst_case_frame_iterations_per_s": (tot2[4] / el2) / worst_fi,
                                    "kernel_ms_avg": kms2 / max(kl2, 1)},
            "roofline_achieved_GBs": (tot2[4] / world * bytes_iter) / (kms2 / 1e3) / 1e9,
        }
        sweep = []
        for snr in np.arange(1.0, 3.01, 0.25):
            s = dec.simulate(float(snr), MAXITER, seed=1, first_frame=rank * B, B=B)
            c = torch.tensor([s["nse"], s["nde"], s["frames"]], dtype=torch.int64, device=dev)
            if world > 1:
                all_reduce(c)
            c = c.cpu().tolist()
            sweep.append({"ebn0_db": float(snr), "fer": c[1] / c[2], "ber": c[0] / c[2] / (N - R), "frames": c[2]})
        out["fer_sweep"] = sweep
    dec.close()

    # ---------------- the other configurations, same run, one GPU (N = 1 only: they are per-GPU figures)
    if not args.no_extras and world == 1:
        from ldpc_testlib import load_base_matrix as lbm
        cfgs = {}
        ksteps = max(3, min(args.steps, 6))
        for c in EXTRA_CONFIGS:
            Hc = relift(lbm(), c["M"])
            try:
                with ldpc_lib_amd.LdpcHip(c["dec"], Hc, c["M"], device=local) as d:
                    bi = algorithmic_bytes_per_iter(c["formula"], d.edges * c["M"], d.R, d.N)
                    e1, t1, k1, l1 = timed(d, c["frames"], WORST_SNR, ksteps, 1, c["maxiter"], c["modulation"])
                    if c["dec"] == DEC_BP:   # the frame chain synchronises the stream after every launch: one batch at a time
                        e2, t2, k2, l2 = timed(d, c["frames"], c["oper_snr"], ksteps, 1, c["maxiter"], c["modulation"])
                    else:                    # like the headline's operating point: two batches in flight on two streams
                        with ldpc_lib_amd.LdpcHip(c["dec"], Hc, c["M"], device=local) as d2:
                            e2, t2, k2, l2 = timed(d, c["frames"], c["oper_snr"], 2 * ksteps, 2, c["maxiter"], c["modulation"], dec_b=d2)
                    cfgs[c["key"]] = {
                        "workload": c["what"], "frames_per_step": c["frames"], "steps": ksteps, "max_iterations": c["maxiter"],
                        "worst_case": {"ebn0_db": WORST_SNR, "value": t1[3] / e1, "unit": "frames/s", "ms_per_step": e1 / ksteps * 1e3,
                                       "mean_iters_per_frame": t1[4] / t1[3], "fer": t1[1] / t1[3]},
                        "operating_point": {"ebn0_db": c["oper_snr"], "value": t2[3] / e2, "unit": "frames/s", "ms_per_step": e2 / max(l2, 1) * 1e3,
                                            "batches_in_flight": 1 if c["dec"] == DEC_BP else 2,
                                            "mean_iters_per_frame": t2[4] / t2[3], "fer": t2[1] / t2[3], "kernel_ms_avg": k2 / max(l2, 1)},
                        "roofline": roofline_block(d.kernel_name, k1 / max(l1, 1), l1, t1[4] / max(l1, 1), c["frames"], bi, pmc, pmc_why, c["key"]),
                    }
            except Exception as ex:
                cfgs[c["key"]] = {"workload": c["what"], "error": repr(ex)}
        out["configs"] = cfgs
        try:
            out["exact_replay"] = exact_replay_block(ldpc_lib_amd, H, local, torch, pmc)
        except Exception as ex:
            out["exact_replay"] = {"error": repr(ex)}

    out["cpu_baseline"] = cpu if (rank == 0 and world == 1) else None

    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank != 0:
        return
    if world == 1 and not args.no_extras:
        try:    # one shard, no communicator: the same entry point the N > 1 leg times
            out["abi_multi"] = abi_multi_leg(1, max(3, min(args.steps, 10)), 1, B)
        except Exception as ex:
            out["abi_multi"] = {"error": repr(ex)}
        out["exact_replay_ranks"] = run_abi_multi_child(args, timeout=480, leg="exact_ranks")   # N = 1: the figures the N > 1 lines must repeat
    elif world > 1 and not args.spawned:
        # started by an external launcher: the other ranks are on their way out; a fresh child process (never this one re-executed)
        # opens all N GPUs behind the C-ABI
        torch.cuda.empty_cache()
        out["abi_multi"] = run_abi_multi_child(args)
        if not args.no_extras:
            out["exact_replay_ranks"] = run_abi_multi_child(args, timeout=480, leg="exact_ranks")
    print(json.dumps(out))


if __name__ == "__main__":
    main()
