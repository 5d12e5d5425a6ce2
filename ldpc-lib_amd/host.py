"""Host-side mirror of the upstream Monte-Carlo harness for the Python callers (bench.py, tests).

`bp_simulation()` keeps upstream's argument meaning, counters, stopping rule and return value
(bp_simulation.h:9-27, bp_simulation.cpp:305-841) but draws the channel noise on the GPU (counter-based Philox keyed by
the global frame index), decodes frames in batches and -- because the upstream stopping rule is sequential in frame
order (bp_simulation.cpp:591,820) -- replays that rule over the ordered per-frame records of each batch, so the result
is exactly what a frame-by-frame loop over the same noise would return, whatever the batch size or GPU count.

`exact_seed=` switches to EXACT REPLAY: the noise is upstream's own stream -- std::mt19937(seed) through a fresh
std::normal_distribution per sample, after the codeword draws of bp_simulation.cpp:512 -- continued on the GPU (csrc/ldpc_mt.hpp), so
counters, BER / FER and the generator state afterwards are upstream's, bit for bit.  With several ranks every rank runs the same
generator over the whole round and decodes its slice.  The C++ layer in csrc/compat/ does the same for upstream's C++ callers.
"""
import os

import numpy as np

from .binding import DEC_IMS, DEC_LMS, DEC_MS, DEC_SP, DEC_TASP, LdpcHip, LdpcHipError

MODULATION_SKIP, MODULATION_QAM4, MODULATION_QAM16 = 0, 1, 2  # modulation.h:4-11


def relift_base_matrix(H, M):
    """main_simulation.cpp:400-414: entries > 0 become entry % M; in column rows-1 a result of 0 becomes 1."""
    H = np.array(H, dtype=np.int32, copy=True)
    rows = H.shape[0]
    pos = H > 0
    H[pos] = H[pos] % M
    col = H[:, rows - 1]
    col[pos[:, rows - 1] & (col == 0)] = 1
    return H


def replay_stopping_rule(frame_info, iters, state, n_frame_errors, n_experiments, reference_frame_error):
    """Apply the frame loop of bp_simulation.cpp:591-823 to one ordered batch of per-frame records.

    frame_info[i]: wrong information bits of frame i, bit 30 set when the frame has any wrong bit.
    state: dict(nse, nde, nue, experiment), updated in place.  Returns True when the simulation stops inside or
    right after this batch.  Error-free frames only advance `experiment`, so the walk visits errored frames only."""
    if not (state["nde"] < n_frame_errors and state["experiment"] <= n_experiments):
        return True
    B = len(frame_info)
    bad = np.flatnonzero((frame_info & (1 << 30)) != 0)
    idx = 0
    for b in bad:
        clean = int(b) - idx
        room = n_experiments + 1 - state["experiment"]  # frames `experiment <= n_experiments` still admits (:591)
        if clean >= room:
            state["experiment"] += room
            return True
        state["experiment"] += clean + 1
        state["nse"] += int(frame_info[b]) & ((1 << 30) - 1)  # :807
        state["nde"] += 1                                      # :808
        if iters[b] >= 0:
            state["nue"] += 1                                  # :809-810
        if state["nde"] >= 10 and state["nde"] / state["experiment"] > 2.5 * reference_frame_error:
            return True                                        # :820
        if state["nde"] >= n_frame_errors:
            return True                                        # :591 fails before the next frame
        idx = int(b) + 1
    clean = B - idx
    room = n_experiments + 1 - state["experiment"]
    if clean >= room:
        state["experiment"] += room
        return True
    state["experiment"] += clean
    return False


class GpuFrameSource:
    """frames(first_frame, B) -> (frame_info[B], iters[B]) int32 tensors: device noise -> decode -> per-frame accounting
    for the global frames [first_frame, first_frame + B).  One instance == one opened code on one GPU."""

    def __init__(self, H, tailbite_length, decoder_type, max_iterations, snr, modulation_type, punctured_blocks, seed,
                 device, alpha):
        self.dec = LdpcHip(decoder_type, np.asarray(H), tailbite_length, device)
        self.n, self.r = self.dec.N, self.dec.R
        self.args = (snr, seed, modulation_type, punctured_blocks, max_iterations, alpha)

    def frames(self, first_frame, B):
        snr, seed, mod, punct, maxit, alpha = self.args
        llr = self.dec.awgn_llr(snr, seed, first_frame, B, modulation=mod, punctured_blocks=punct)
        hard, iters, _ = self.dec.decode(llr, maxit, alpha=alpha)
        _, info = self.dec.count_errors(hard, iters, want_frame_info=True)
        return info, iters

    def close(self):
        self.dec.close()


def mt19937_state(seed, skip_words=0):
    """(624 words, next index) of std::mt19937(seed) after `skip_words` raw draws: init_genrand, then the public recurrence
    (numpy's MT19937 bit generator walks it)."""
    key = np.empty(624, dtype=np.uint64)
    key[0] = int(seed) & 0xffffffff
    for i in range(1, 624):
        key[i] = (1812433253 * (int(key[i - 1]) ^ (int(key[i - 1]) >> 30)) + i) & 0xffffffff
    bg = np.random.MT19937()
    st = bg.state
    st["state"]["key"] = key.astype(np.uint32)
    st["state"]["pos"] = 624
    bg.state = st
    if skip_words:
        bg.random_raw(int(skip_words))
    return bg.state["state"]["key"].astype(np.uint32), int(bg.state["state"]["pos"])


class MtFrameSource:
    """Exact replay: round(total, lo, hi) -> records of frames [lo, hi) of the next `total` frames of upstream's own noise stream
    (ldpc_hip_mt_frames_slice).  Every rank holds the same generator and advances it by `total`."""

    def __init__(self, H, tailbite_length, decoder_type, max_iterations, snr, modulation_type, punctured_blocks, seed, device, alpha):
        H = np.asarray(H)
        self.dec = LdpcHip(decoder_type, H, tailbite_length, device)
        self.n, self.r = self.dec.N, self.dec.R
        self.args = (snr, modulation_type, punctured_blocks, max_iterations, alpha)
        self.share_tape = os.environ.get("LDPC_HIP_MT_SHARDED", "1") != "0"
        self.shared_rounds = self.fallback_rounds = 0
        # random_codeword() draws (nh - rh) * M values of next_random_int(0, 2) first, one generator word each (bp_simulation.cpp:512,160-162)
        self.dec.mt_set_state(*mt19937_state(seed, (H.shape[1] - H.shape[0]) * tailbite_length))

    def snapshot(self):
        """(624 words, next index, frames drawn so far): the frame count is part of the snapshot (codeword f % ncw of frame f)"""
        st, pos = self.dec.mt_get_state()
        return st, pos, self.dec.mt_frame_index()

    def restore_and_skip(self, snap, frames):
        """the generator where a frame-by-frame loop that stopped after `frames` frames of the round would have left it"""
        snr, mod, punct, _, _ = self.args
        self.dec.mt_set_state(snap[0], snap[1])
        self.dec.mt_set_frame_index(snap[2])
        if frames:
            self.dec.mt_llr(snr, frames, modulation=mod, punctured_blocks=punct, skip=True)

    group = None   # torch.distributed group of the job (bp_simulation sets it)
    share_single = False   # tests: run the exchange protocol in a job of ONE rank too (RCCL collectives on the one-GPU box)

    def round(self, total, lo, hi):
        """records of frames [lo, hi) of the next `total` frames.  In a torch.distributed job with equal contiguous slices the ranks
        share the generator's tape out (ldpc_hip_mt_shard_*: every rank makes ~1/n of the words; three small exchanges per round);
        otherwise -- one rank, a round beyond 65536 frames, or an estimate that failed -- every rank runs the whole generator."""
        import torch
        import torch.distributed as dist
        snr, mod, punct, maxit, alpha = self.args
        group = self.group
        # device tensors: the record exchange may run over RCCL (a decoder without a device: the CPU tests' stand-in)
        dev = torch.device("cuda", self.dec.device) if self.dec.device is not None else torch.device("cpu")
        world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
        rank = dist.get_rank(group) if world > 1 else 0
        in_job = dist.is_available() and dist.is_initialized()
        shared = ((world > 1 or (self.share_single and in_job)) and self.share_tape and total <= 65536 and total * self.n <= (1 << 27)
                  and lo == total * rank // world and hi == total * (rank + 1) // world)
        if shared:
            host = torch.device("cpu")

            def gather(vals):   # a few integers per rank, through the host (gloo) or the device (RCCL)
                use_dev = dist.get_backend(group) == "nccl"
                t = torch.tensor(vals, dtype=torch.int64, device=dev if use_dev else host)
                out = [torch.empty_like(t) for _ in range(world)]
                dist.all_gather(out, t, group=group)
                return [o.cpu().tolist() for o in out]

            own = self.dec.mt_shard_begin(snr, total, rank, world, modulation=mod, punctured_blocks=punct)
            counts = [g[0] for g in gather([own])]
            found, covered, st, fdone = self.dec.mt_shard_emit(counts)
            flags = gather([int(found), int(covered)])
            owner = next((r for r, f in enumerate(flags) if f[0]), None)
            if owner is not None and all(f[1] for f in flags):
                use_dev = dist.get_backend(group) == "nccl"
                t = torch.from_numpy(st.astype(np.int64)).to(dev if use_dev else host)
                dist.broadcast(t, src=dist.get_global_rank(group, owner) if group is not None else owner, group=group)
                state = t.cpu().numpy().astype(np.uint32)
                rows = max(min(hi, fdone) - lo, 0)
                info, its = self.dec.mt_shard_commit(state, fdone, maxit, rows, alpha=alpha)
                self.shared_rounds += 1
                if fdone == total:
                    return torch.from_numpy(info).to(dev), torch.from_numpy(its).to(dev)
                # a short round (the number of accepted attempts is random; 8 sigma): the rest of the frames the plain way
                i2, t2 = self.dec.mt_frames(snr, maxit, total - fdone, modulation=mod, punctured_blocks=punct, alpha=alpha,
                                            lo=max(lo - fdone, 0), hi=max(hi - fdone, 0))
                return torch.from_numpy(np.concatenate([info, i2])).to(dev), torch.from_numpy(np.concatenate([its, t2])).to(dev)
            self.dec.mt_shard_abandon()
            self.fallback_rounds += 1
        info, its = self.dec.mt_frames(snr, maxit, total, modulation=mod, punctured_blocks=punct, alpha=alpha, lo=lo, hi=hi)
        return torch.from_numpy(info).to(dev), torch.from_numpy(its).to(dev)

    def close(self):
        self.dec.close()


def bp_simulation(H, tailbite_length, max_iterations, n_frame_errors, n_experiments, snr, reference_frame_error,
                  decoder_type=DEC_MS, modulation_type=MODULATION_SKIP, punctured_blocks=0, seed=1, device=0,
                  batch=16384, alpha=0.8, return_state=False, source=None, group=None, exact_seed=None):
    """Returns (BER, FER) = (nse/experiment/(n-r), nde/experiment) like bp_simulation.cpp:840.

    Multi-GPU (torch.distributed initialised, one process per GPU): every round covers world*batch consecutive global
    frames, rank r decodes the r-th slice, the 8-byte per-frame records are all-gathered (no other collective), and
    every rank replays the sequential stopping rule over the round in global frame order -- so all ranks take the same
    decisions and the result equals the single-GPU (and the frame-by-frame) result for the same seed.
    `source` (tests): any object with frames(first, B) -> (info, iters) tensors, n, r, close()."""
    import torch
    import torch.distributed as dist
    world, rank = 1, 0
    if dist.is_available() and dist.is_initialized():
        world, rank = dist.get_world_size(group), dist.get_rank(group)
    if source is None and exact_seed is not None:
        src = MtFrameSource(H, tailbite_length, decoder_type, max_iterations, snr, modulation_type, punctured_blocks, exact_seed, device, alpha)
    else:
        src = source if source is not None else GpuFrameSource(H, tailbite_length, decoder_type, max_iterations, snr,
                                                               modulation_type, punctured_blocks, seed, device, alpha)
    exact = isinstance(src, MtFrameSource)
    if exact:
        src.group = group
    n, r = src.n, src.r
    state = {"nse": 0, "nde": 0, "nue": 0, "experiment": 0, "sum_abs_iters": 0}
    base, stop = 0, False
    try:
        while not stop:
            room = max(1, n_experiments + 1 - state["experiment"])
            B = int(min(batch, -(-room // world)))           # frames per rank this round
            if exact:
                snap = src.snapshot()
                info, iters = src.round(world * B, rank * B, (rank + 1) * B)
            else:
                info, iters = src.frames(base + rank * B, B)
            rec = torch.stack([info.to(torch.int32), iters.to(torch.int32)])  # [2, B]
            if world > 1:
                gathered = [torch.empty_like(rec) for _ in range(world)]
                dist.all_gather(gathered, rec, group=group)
                rec = torch.cat(gathered, dim=1)             # global frame order: rank-major slices are consecutive
            rec_h = rec.cpu().numpy()
            before = state["experiment"]
            stop = replay_stopping_rule(rec_h[0], rec_h[1], state, n_frame_errors, n_experiments, reference_frame_error)
            used = state["experiment"] - before
            state["sum_abs_iters"] += int(np.abs(rec_h[1][:used]).sum())
            if exact and used < world * B:
                src.restore_and_skip(snap, used)      # stopped inside the round
            base += world * B
        if exact:
            snap = src.snapshot()                     # (624 words, next index): what upstream's `generator` holds afterwards
            state["generator"] = snap[:2] if isinstance(snap, tuple) else snap
            state["tape_shared_rounds"] = getattr(src, "shared_rounds", 0)
            state["tape_fallback_rounds"] = getattr(src, "fallback_rounds", 0)
    finally:
        if source is None:
            src.close()
    ber = state["nse"] / state["experiment"] / (n - r)
    fer = state["nde"] / state["experiment"]
    return (ber, fer, state) if return_state else (ber, fer)
