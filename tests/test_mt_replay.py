"""Upstream's noise stream on the device (ldpc_hip_mt_*, csrc/ldpc_mt.hpp): std::mt19937 + a fresh std::normal_distribution<double>
per sample (commons_portable.cpp:140,174-178), continued on the GPU bit for bit.

CPU part: the GF(2) jump polynomials against a plain walk of the recurrence (numpy's MT19937 bit generator is the same public
algorithm).  GPU part: samples, LLR rows and generator state against the oracle's std:: objects (oracle/harness_oracle.cpp), and the
C++ harness with the noise on the device against the sequential restatement of upstream's frame loop."""
import ctypes as C
import os

import numpy as np
import pytest

from ldpc_testlib import MS_DEC, LMS_DEC, SP_DEC, TASP_DEC, BP_DEC, load_base_matrix, oracle_lib, relift, awgn_llr, code_rate


def seeded_state(seed):
    """std::mt19937(seed): init_genrand's 624 words, position 624 (the first draw regenerates the block)."""
    bg = np.random.MT19937()
    bg._legacy_seeding(int(seed))
    st = bg.state["state"]
    return st["key"].astype(np.uint32).copy(), int(st["pos"])


def raw_after(key, pos, skip, n):
    """tempered outputs skip .. skip+n of the generator (key, pos)"""
    bg = np.random.MT19937()
    s = bg.state
    s["state"]["key"] = np.array(key, dtype=np.uint32)
    s["state"]["pos"] = int(pos)
    bg.state = s
    if skip:
        bg.random_raw(int(skip))
    return bg.random_raw(int(n)).astype(np.uint32)


def oracle_gaussians(seed, n, burn=0):
    out = np.empty(n, dtype=np.float64)
    oracle_lib().orc_rng_gaussians(int(seed), int(burn), out.ctypes.data_as(C.POINTER(C.c_double)), int(n))
    return out


@pytest.fixture(scope="module")
def L():
    import ldpc_lib_amd
    ldpc_lib_amd.build_library()
    return ldpc_lib_amd


@pytest.mark.parametrize("log2_words", [18, 19, 20, 21, 24, 26])
def test_jump_polynomial_equals_a_plain_walk(L, log2_words):
    """state after 2^k words by x^(2^k) mod phi == the recurrence walked 2^k words (numpy's MT19937)."""
    from ldpc_lib_amd.binding import mt_jump_host
    key, _ = seeded_state(20240607 + log2_words)
    jumped = mt_jump_host(key, log2_words)
    J = 1 << log2_words
    # sequence origin: x[0..624) = key; outputs from position 624 on are temper(x[624 + i])
    want = raw_after(key, 624, J, 1500)            # temper(x[J + 624 + i])
    got = raw_after(jumped, 624, 0, 1500)          # the same words from the jumped state (word 0: only bit 31 is state)
    assert np.array_equal(want, got)
    # words 1..623 of the jumped state are x[J + 1 ..]: temper them and compare with the walk's outputs J - 623 .. J
    direct = raw_after(key, 624, J - 623, 623)     # temper(x[J + 1 .. J + 623])
    assert np.array_equal(direct, raw_after(jumped, 1, 0, 623))


def test_jump_levels_compose(L):
    """jump(2^29) == jump(2^28) twice (the longest stride is too long to walk in a test)."""
    from ldpc_lib_amd.binding import mt_jump_host
    key, _ = seeded_state(77)
    once = mt_jump_host(key, 29)
    twice = mt_jump_host(mt_jump_host(key, 28), 28)
    assert np.array_equal(once[1:], twice[1:]) and (once[0] ^ twice[0]) & 0x80000000 == 0


# ---- GPU ------------------------------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def torch():
    import torch as t
    if not t.cuda.is_available():
        pytest.skip("no GPU")
    return t


@pytest.mark.gpu
@pytest.mark.parametrize("seed,burn,counts", [
    (1, 0, [1, 7, 4096, 200000]),          # single stream, uneven pieces: every call continues where the last one stopped
    (5, 1024, [3_000_000]),                # 15 streams: jump levels 0..3; the generator starts inside its block (position 400)
    (9, 0, [30_000_000, 1000]),            # 146 streams, levels 0..7, then a short continuation
])
def test_device_samples_equal_the_hosts(L, torch, seed, burn, counts):
    """next_random_gaussian() on the device == std::normal_distribution on the host, bit for bit, whatever the call split."""
    H = relift(load_base_matrix(), 64)
    total = sum(counts)
    want = oracle_gaussians(seed, total, burn)
    key, pos = seeded_state(seed)
    if burn:   # next_random_int(0, 2) takes one word per draw: start inside the first regenerated block
        bg = np.random.MT19937()
        s = bg.state
        s["state"]["key"] = key; s["state"]["pos"] = pos
        bg.state = s
        bg.random_raw(burn)
        key, pos = bg.state["state"]["key"].astype(np.uint32), int(bg.state["state"]["pos"])
    with L.LdpcHip(MS_DEC, H, 64) as dec:
        dec.mt_set_state(key, pos)
        got = torch.cat([dec.mt_normal(c) for c in counts]).cpu().numpy()
        assert np.array_equal(got.view(np.uint64), want.view(np.uint64))
        # the state handed back continues the host stream: its next words are what the host generator would draw next
        st, p = dec.mt_get_state()
        nxt = raw_after(st, p, 0, 2000)
        # host: replay the same number of samples with the oracle and look at the values after them
        more = oracle_gaussians(seed, total + 50, burn)[total:]
        dec.mt_set_state(st, p)
        assert np.array_equal(dec.mt_normal(50).cpu().numpy().view(np.uint64), more.view(np.uint64))
        assert nxt.shape == (2000,)


@pytest.mark.gpu
def test_skipping_equals_drawing(L, torch):
    H = relift(load_base_matrix(), 64)
    key, pos = seeded_state(3)
    with L.LdpcHip(MS_DEC, H, 64) as dec:
        dec.mt_set_state(key, pos)
        dec.mt_normal(123457, skip=True)
        a = dec.mt_normal(1000).cpu().numpy()
        want = oracle_gaussians(3, 123457 + 1000)[123457:]
        assert np.array_equal(a.view(np.uint64), want.view(np.uint64))


@pytest.mark.gpu
@pytest.mark.parametrize("M,snr,frames,seed", [(64, 2.0, 300, 1), (1, 4.0, 5000, 2), (126, 1.7, 40, 3)])
def test_device_llr_rows_equal_the_frame_loops(L, torch, M, snr, frames, seed):
    """bp_simulation.cpp:512,600-605: codeword draws first, then N samples per frame -> y = -2 (sigma g - 1) / sigma^2."""
    H = relift(load_base_matrix(), M)
    want = awgn_llr(H, M, snr, seed, frames)
    key, pos = seeded_state(seed)
    bg = np.random.MT19937()
    s = bg.state
    s["state"]["key"] = key; s["state"]["pos"] = pos
    bg.state = s
    bg.random_raw((H.shape[1] - H.shape[0]) * M)       # random_codeword()'s next_random_int(0, 2) draws, one word each
    key, pos = bg.state["state"]["key"].astype(np.uint32), int(bg.state["state"]["pos"])
    with L.LdpcHip(MS_DEC, H, M) as dec:
        dec.mt_set_state(key, pos)
        first = dec.mt_llr(snr, frames // 3).cpu().numpy()
        rest = dec.mt_llr(snr, frames - frames // 3).cpu().numpy()
        got = np.concatenate([first, rest])
        assert np.array_equal(got.view(np.uint64), want.view(np.uint64))


def _compat(L):
    import subprocess
    L.load_library()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    subprocess.check_call(["make", "-s", "-C", os.path.join(root, "ldpc-lib_amd", "csrc", "compat")])
    lib = C.CDLL(os.path.join(root, "ldpc-lib_amd", "libldpc_compat.so"))
    lib.ldpc_bp_simulation_exact_perm.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, C.c_double, C.c_int,
                                                  C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_uint, C.c_int, C.c_void_p, C.c_void_p]
    return lib


@pytest.mark.gpu
@pytest.mark.parametrize("noise", ["device", "host"])
@pytest.mark.parametrize("dec_id,M,snr,maxit,n_fe,n_exp,ref,mod,punct,seed,devices", [
    (MS_DEC, 64, 2.0, 50, 10**9, 4000, 1.0, 0, 0, 1, "0"),          # cfg2: 170 / 4001
    (MS_DEC, 64, 1.2, 50, 10**9, 5000, 0.02, 0, 0, 1, "0"),         # early stop on the FER rule: roll-back inside a batch
    (MS_DEC, 64, 1.4, 50, 7, 5000, 1.0, 0, 0, 3, "0,0,0"),          # stops on n_frame_errors; three logical shards
    (LMS_DEC, 64, 1.6, 50, 10**9, 300, 1.0, 1, 2, 9, "0,0"),        # QAM4 + two punctured blocks, two shards
    (TASP_DEC, 126, 1.7, 15, 50, 10**8, 1.0, 0, 0, 1, "0"),         # the shipped scenario: 50 / 821
    (BP_DEC, 64, 1.3, 30, 10**9, 250, 1.0, 0, 0, 2, "0,0"),         # frame chain: shard 0 decodes, both generators advance
    (SP_DEC, 64, 1.9, 50, 10**9, 40000, 1.0, 0, 0, 4, "0"),         # a run long enough for the 32768-frame batches
])
def test_harness_with_the_noise_on_the_device(L, torch, monkeypatch, noise, dec_id, M, snr, maxit, n_fe, n_exp, ref, mod, punct, seed, devices):
    """ldpc::bp_simulation_t with LDPC_HIP_EXACT_NOISE=device (default) / host == the sequential CPU restatement: counters, BER / FER
    doubles and the next word of the generator afterwards."""
    from ldpc_testlib import SimResult, c_int_p
    if noise == "host" and n_exp > 6000:
        pytest.skip("the host-noise harness is the slow one; the long run is for the device path")
    monkeypatch.setenv("LDPC_HIP_EXACT_NOISE", noise)
    monkeypatch.setenv("LDPC_HIP_DEVICES", devices)
    lib = _compat(L)
    H = np.ascontiguousarray(relift(load_base_matrix(), M), dtype=np.int32)
    out = (C.c_double * 7)()
    nxt = C.c_uint()
    rc = lib.ldpc_bp_simulation_exact_perm(16, 32, H.ctypes.data, M, maxit, n_fe, n_exp, snr, ref, dec_id, mod, 0, 128, 1, punct, seed, 0,
                                           C.addressof(out), C.addressof(nxt))
    assert rc == 0
    res = SimResult()
    assert oracle_lib().orc_bp_simulation(16, 32, H.ctypes.data_as(c_int_p), M, maxit, n_fe, n_exp, snr, ref, dec_id, mod, punct,
                                          seed, C.byref(res), None) == 0
    assert (out[2], out[3], out[4], out[5], out[6]) == (res.nse, res.nde, res.nue, res.experiment, res.sum_abs_iter)
    assert out[0] == res.ber and out[1] == res.fer
    assert nxt.value == res.rng_next
    if (dec_id, n_exp, seed) == (MS_DEC, 4000, 1):
        assert (res.nde, res.experiment) == (170, 4001)


@pytest.mark.gpu
@pytest.mark.parametrize("perm_type,block,inter", [(3, 64, 1), (1, 128, 1)])
def test_harness_on_the_device_with_an_interleaver(L, torch, monkeypatch, perm_type, block, inter):
    from ldpc_testlib import SimResult, c_int_p
    from ldpc_lib_amd.binding import build_interleaver
    monkeypatch.setenv("LDPC_HIP_EXACT_NOISE", "device")
    lib = _compat(L)
    M = 64
    H = np.ascontiguousarray(relift(load_base_matrix(), M), dtype=np.int32)
    out = (C.c_double * 7)()
    nxt = C.c_uint()
    assert lib.ldpc_bp_simulation_exact_perm(16, 32, H.ctypes.data, M, 50, 10**9, 400, 1.8, 1.0, MS_DEC, 0, perm_type, block, inter, 0, 11, 0,
                                             C.addressof(out), C.addressof(nxt)) == 0
    _, inv = build_interleaver(H, M, perm_type, 1, block, inter)
    inv = np.ascontiguousarray(inv, dtype=np.int32)
    olib = oracle_lib()
    olib.orc_bp_simulation_perm.argtypes = [C.c_int, C.c_int, c_int_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, C.c_double, C.c_int, C.c_int,
                                            C.c_int, C.c_uint, c_int_p, C.POINTER(SimResult), c_int_p]
    res = SimResult()
    assert olib.orc_bp_simulation_perm(16, 32, H.ctypes.data_as(c_int_p), M, 50, 10**9, 400, 1.8, 1.0, MS_DEC, 0, 0, 11,
                                       inv.ctypes.data_as(c_int_p), C.byref(res), None) == 0
    assert (out[2], out[3], out[4], out[5]) == (res.nse, res.nde, res.nue, res.experiment)
    assert out[0] == res.ber and out[1] == res.fer and nxt.value == res.rng_next


@pytest.mark.gpu
@pytest.mark.parametrize("modulation,punct,perm,with_codewords,dec_id", [
    (0, 0, 0, False, SP_DEC),      # probability-type decoder: nothing punctured, plain
    (1, 2, 3, True, MS_DEC),       # QAM4 sigma (:449), two punctured blocks (0.5, :700), block interleaver, real codewords
    (0, 1, 1, True, TASP_DEC),     # out_type 1 puncturing value 0
])
def test_device_llr_chain_options(L, torch, modulation, punct, perm, with_codewords, dec_id):
    """mt_llr with every option of the frame loop == the same chain composed in numpy from the host's samples:
    y[i] = buffer[inverse[i]], buffer[j] = -2 (sigma g_j + 2 c_j - 1) / sigma^2, c = codeword through the direct map, tail punctured."""
    from ldpc_lib_amd.binding import build_interleaver, encode
    M, frames, seed, snr = 64, 50, 12, 2.2
    H = relift(load_base_matrix(), M)
    rh, nh = H.shape
    N = nh * M
    g = oracle_gaussians(seed, frames * N).reshape(frames, N)
    bitrate = (nh - rh) / (nh - punct)
    if modulation == 0:
        sigma = np.sqrt(np.power(10.0, -snr / 10) / 2 / bitrate)
    else:
        sigma = np.sqrt(np.power(10.0, -snr / 10.0) / (2 * bitrate * 1 * 2) * (2.0 * (4 - 1.0) / 3.0))
    rng = np.random.default_rng(5)
    ncw = 3
    cws = np.stack([encode(H, M, rng.integers(0, 2, (nh - rh) * M, dtype=np.uint8)) for _ in range(ncw)]) if with_codewords else None
    direct = inverse = np.arange(N)
    if perm:
        direct, inverse = build_interleaver(H, M, perm, 1, 64, 1)
    want = np.empty((frames, N))
    for f in range(frames):
        c = cws[f % ncw][direct].astype(np.float64) if with_codewords else np.zeros(N)
        buf = -2.0 * (sigma * g[f] + 2.0 * c - 1.0) / (sigma * sigma)
        y = buf[inverse]
        if punct:
            y[N - M * punct:] = 0.0 if dec_id in (SP_DEC, TASP_DEC) else 0.5
        want[f] = y
    key, pos = seeded_state(seed)
    with L.LdpcHip(dec_id, H, M) as dec:
        if perm:
            dec.set_interleaver(perm, 64, 1)
        if with_codewords:
            dec.set_codewords(cws)
        dec.mt_set_state(key, pos)
        got = torch.cat([dec.mt_llr(snr, 20, modulation=modulation, punctured_blocks=punct),
                         dec.mt_llr(snr, frames - 20, modulation=modulation, punctured_blocks=punct)]).cpu().numpy()
    assert np.array_equal(got.view(np.uint64), want.view(np.uint64))


@pytest.mark.gpu
def test_device_generator_soak(L, torch):
    """random seeds, start positions inside the block and call splits (single words up to several streams)"""
    H = relift(load_base_matrix(), 1)
    rng = np.random.default_rng(2024)
    with L.LdpcHip(MS_DEC, H, 1) as dec:
        for case in range(12):
            seed = int(rng.integers(1, 2**31))
            burn = int(rng.integers(0, 3000))
            counts = [int(c) for c in rng.choice([1, 2, 3, 63, 64, 65, 1000, 4097, 300_000, 1_500_000], size=int(rng.integers(1, 6)))]
            want = oracle_gaussians(seed, sum(counts), burn)
            key, pos = seeded_state(seed)
            if burn:
                bg = np.random.MT19937()
                s = bg.state
                s["state"]["key"] = key; s["state"]["pos"] = pos
                bg.state = s
                bg.random_raw(burn)
                key, pos = bg.state["state"]["key"].astype(np.uint32), int(bg.state["state"]["pos"])
            dec.mt_set_state(key, pos)
            got = torch.cat([dec.mt_normal(c) for c in counts]).cpu().numpy()
            assert np.array_equal(got.view(np.uint64), want.view(np.uint64)), (case, seed, burn, counts)


@pytest.mark.gpu
def test_generator_needs_a_state_and_sane_arguments(L, torch):
    H = relift(load_base_matrix(), 64)
    with L.LdpcHip(MS_DEC, H, 64) as dec:
        with pytest.raises(L.LdpcHipError):
            dec.mt_normal(10)                       # no state loaded
        key, pos = seeded_state(1)
        with pytest.raises(L.LdpcHipError):
            dec.mt_set_state(key, 625)
        dec.mt_set_state(key, pos)
        with pytest.raises(L.LdpcHipError):
            dec.mt_llr(2.0, 4, modulation=2)        # QAM16+: nothing to replay
        assert dec.mt_normal(0).numel() == 0
        info, its = dec.mt_frames(2.0, 50, 0)
        assert info.shape == (0,) and its.shape == (0,)
        # the rank protocol: nothing to emit or commit without a round, no round from bad arguments -- and none left behind by them
        for bad in (lambda: dec.mt_shard_emit([0]), lambda: dec.mt_shard_commit(np.zeros(624, dtype=np.uint32), 0, 50, 0),
                    lambda: dec.mt_shard_begin(2.0, 10, 3, 2), lambda: dec.mt_shard_begin(2.0, 70000, 0, 2), lambda: dec.mt_shard_begin(2.0, 0, 0, 1),
                    lambda: dec.mt_shard_emit([0, 0])):
            with pytest.raises(L.LdpcHipError):
                bad()


# ---- the Python harness (ldpc_lib_amd.bp_simulation(exact_seed=...)), one process and two ranks ----------------------------------
_EXACT_CASES = [(MS_DEC, 64, 2.0, 50, 10**9, 4000, 1.0, 1), (MS_DEC, 64, 1.2, 50, 10**9, 5000, 0.02, 1), (LMS_DEC, 64, 1.5, 50, 12, 10**6, 1.0, 7)]


def _oracle_sim(H, M, dec_id, snr, maxit, n_fe, n_exp, ref, seed):
    from ldpc_testlib import SimResult, c_int_p
    Hc = np.ascontiguousarray(H, dtype=np.int32)
    res = SimResult()
    assert oracle_lib().orc_bp_simulation(16, 32, Hc.ctypes.data_as(c_int_p), M, maxit, n_fe, n_exp, snr, ref, dec_id, 0, 0, seed, C.byref(res), None) == 0
    return res


@pytest.mark.gpu
@pytest.mark.parametrize("dec_id,M,snr,maxit,n_fe,n_exp,ref,seed", _EXACT_CASES)
def test_python_harness_exact_replay(L, torch, dec_id, M, snr, maxit, n_fe, n_exp, ref, seed):
    H = relift(load_base_matrix(), M)
    ber, fer, st = L.bp_simulation(H, M, maxit, n_fe, n_exp, snr, ref, decoder_type=dec_id, exact_seed=seed, batch=3000, return_state=True)
    res = _oracle_sim(H, M, dec_id, snr, maxit, n_fe, n_exp, ref, seed)
    assert (st["nse"], st["nde"], st["nue"], st["experiment"], st["sum_abs_iters"]) == (res.nse, res.nde, res.nue, res.experiment, res.sum_abs_iter)
    assert ber == res.ber and fer == res.fer
    assert int(raw_after(*st["generator"], 0, 1)[0]) == res.rng_next


_EXACT_RANK_WORKER = r"""
import os, sys, json
sys.path.insert(0, {root!r}); sys.path.insert(0, os.path.join({root!r}, "tests"))
import numpy as np, torch, torch.distributed as dist
import ldpc_lib_amd
from ldpc_testlib import load_base_matrix, relift
from test_mt_replay import _EXACT_CASES, raw_after
dist.init_process_group("gloo", rank=int(os.environ["RANK"]), world_size=int(os.environ["WORLD_SIZE"]))
out, shared = [], []
for dec_id, M, snr, maxit, n_fe, n_exp, ref, seed in _EXACT_CASES:
    H = relift(load_base_matrix(), M)
    ber, fer, st = ldpc_lib_amd.bp_simulation(H, M, maxit, n_fe, n_exp, snr, ref, decoder_type=dec_id, exact_seed=seed, batch=700, return_state=True)
    out.append([st["nse"], st["nde"], st["nue"], st["experiment"], st["sum_abs_iters"], ber.hex(), fer.hex(), int(raw_after(*st["generator"], 0, 1)[0])])
    shared.append([st["tape_shared_rounds"], st["tape_fallback_rounds"]])
print("RESULT", dist.get_rank(), json.dumps(out))
print("SHARED", dist.get_rank(), json.dumps(shared))
dist.destroy_process_group()
"""


@pytest.mark.gpu
@pytest.mark.parametrize("world,share", [(2, "1"), (3, "1"), (2, "0")])
def test_two_ranks_exact_replay(L, torch, world, share):
    """one process per GPU (here: two or three processes on the one GPU, gloo for the exchanges): the ranks share the generator's tape
    out (ldpc_hip_mt_shard_begin / emit / commit: counts all-gathered, the end state broadcast) or -- LDPC_HIP_MT_SHARDED=0 -- every
    rank runs the whole generator; either way every rank must report the sequential loop's counters, doubles and generator state,
    and in the first case every round must have run shared, none fallen back."""
    import json
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    want = []
    for dec_id, M, snr, maxit, n_fe, n_exp, ref, seed in _EXACT_CASES:
        res = _oracle_sim(relift(load_base_matrix(), M), M, dec_id, snr, maxit, n_fe, n_exp, ref, seed)
        want.append([res.nse, res.nde, res.nue, res.experiment, res.sum_abs_iter, float(res.ber).hex(), float(res.fer).hex(), res.rng_next])
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0",
                   LDPC_HIP_MT_SHARDED=share)
        procs.append(subprocess.Popen([sys.executable, "-c", _EXACT_RANK_WORKER.format(root=root)], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=600)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    for o in outs:
        line = [ln for ln in o.splitlines() if ln.startswith("RESULT")][0]
        assert json.loads(line.split(" ", 2)[2]) == want
        sh = json.loads([ln for ln in o.splitlines() if ln.startswith("SHARED")][0].split(" ", 2)[2])
        if share == "1":
            assert all(a > 0 and b == 0 for a, b in sh), sh
        else:
            assert all(a == 0 for a, b in sh), sh


@pytest.mark.gpu
def test_c_example_replays_the_headline_run(L, torch, tmp_path):
    """examples/exact_replay.c: the exact-replay entry points from plain C -- cfg2, seed 1: 170 errored frames in 4001."""
    import subprocess
    from ldpc_testlib import GOLDEN_DIR
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "exact_replay")
    subprocess.check_call(["gcc", "-O1", "-Wall", "-Werror", "-I", os.path.join(root, "include"), os.path.join(root, "examples", "exact_replay.c"), "-o", exe,
                           "-L", os.path.join(root, "ldpc-lib_amd"), "-lldpc_hip", "-Wl,-rpath," + os.path.join(root, "ldpc-lib_amd")])
    out = subprocess.check_output([exe, os.path.join(GOLDEN_DIR, "h16x32_m126.txt"), "64", "3", "50", "2.0", "4000", "1"]).decode().split()
    assert out[out.index("frames") + 1] == "4001" and out[out.index("errored") + 1] == "170", out
    res = _oracle_sim(relift(load_base_matrix(), 64), 64, MS_DEC, 2.0, 50, 10**9, 4000, 1.0, 1)
    # the state the example reports continues upstream's stream: its next word is the sequential loop's next word
    assert int(out[out.index("undetected") + 1]) == res.nue


# ---- the generator shared out over the shards of a multi context (csrc/ldpc_multi.hpp: multi_mt_round) ---------------------------------
def _after_codeword_draws(H, M, seed):
    key, pos = seeded_state(seed)
    bg = np.random.MT19937()
    s = bg.state
    s["state"]["key"] = key; s["state"]["pos"] = pos
    bg.state = s
    bg.random_raw((H.shape[1] - H.shape[0]) * M)
    return bg.state["state"]["key"].astype(np.uint32), int(bg.state["state"]["pos"])


@pytest.mark.gpu
@pytest.mark.parametrize("forced", ["1", None])
@pytest.mark.parametrize("dec_id,M,snr,frames,splits", [(MS_DEC, 64, 2.0, 9000, (4000, 5000)), (LMS_DEC, 126, 1.7, 2500, (2500,)), (MS_DEC, 1, 4.0, 70000, (69000, 1000)),
                                                         (BP_DEC, 64, 2.0, 700, (300, 400))])
def test_sharded_generation_gives_the_single_context_records_and_state(L, torch, monkeypatch, dec_id, M, snr, frames, splits, forced):
    """ldpc_hip_mt_frames_multi with the word tape shared out (every shard makes ~1/n of the sub-streams, the accepted-attempt counts
    are exchanged through the host): per-frame records in global order and the generator state afterwards equal the single-context
    run for n = 1, 2, 3, 8 logical shards, in several calls; with LDPC_HIP_MT_SHARDED=1 every round must have run sharded and none
    may have fallen back to the whole tape.  BP_DEC (frame chain on) decodes on shard 0 while the generation is still shared."""
    if forced:
        monkeypatch.setenv("LDPC_HIP_MT_SHARDED", forced)
    H = relift(load_base_matrix(), M)
    key, pos = _after_codeword_draws(H, M, 11)
    maxit = 20
    with L.LdpcHip(dec_id, H, M) as dec:
        dec.mt_set_state(key, pos)
        parts = [dec.mt_frames(snr, maxit, b) for b in splits]
        want_info, want_it = np.concatenate([p[0] for p in parts]), np.concatenate([p[1] for p in parts])
        want_state = dec.mt_get_state()
        assert dec.mt_frame_index() == frames
    assert (want_info != 0).any()
    for n in (1, 2, 3, 8):
        with L.LdpcHipMulti(dec_id, H, M, [0] * n) as m:
            m.mt_set_state(key, pos)
            parts = [m.mt_frames(snr, maxit, b) for b in splits]
            info, its = np.concatenate([p[0] for p in parts]), np.concatenate([p[1] for p in parts])
            st = m.mt_get_state()
            sharded, fallback = m.mt_stats()
            assert np.array_equal(info, want_info) and np.array_equal(its, want_it), n
            assert np.array_equal(st[0], want_state[0]) and st[1] == want_state[1], n
            if forced and n > 1:
                assert sharded >= len(splits) and fallback == 0, (n, sharded, fallback)
            if n == 1:
                assert sharded == 0


@pytest.mark.gpu
def test_sharded_generation_with_the_whole_chain_and_roll_forward(L, torch, monkeypatch):
    """QAM4 + block interleaver + two punctured blocks + several real codewords through the shared-out generator; then a roll-back
    as the harness does it after an early stop (set_state + frame index + advance): same records for every shard count."""
    monkeypatch.setenv("LDPC_HIP_MT_SHARDED", "1")
    M = 64
    H = relift(load_base_matrix(), M)
    key, pos = _after_codeword_draws(H, M, 5)
    from ldpc_lib_amd.binding import encode
    rng = np.random.RandomState(3)
    cws = np.stack([encode(H, M, rng.randint(0, 2, size=(H.shape[1] - H.shape[0]) * M).astype(np.uint8)) for _ in range(3)])

    def run(obj):
        obj.set_interleaver(3, 64, 1)
        obj.set_codewords(cws)
        obj.mt_set_state(key, pos)
        a = obj.mt_frames(1.6, 30, 3000, modulation=1, punctured_blocks=2)
        snap = obj.mt_get_state()
        b = obj.mt_frames(1.6, 30, 2000, modulation=1, punctured_blocks=2)
        obj.mt_set_state(*snap)                                        # stopped 700 frames into the second call
        obj.mt_set_frame_index(3000)
        if hasattr(obj, "mt_advance"):
            obj.mt_advance(1.6, 700, modulation=1, punctured_blocks=2)
        else:
            obj.mt_llr(1.6, 700, modulation=1, punctured_blocks=2, skip=True)
        c = obj.mt_frames(1.6, 30, 500, modulation=1, punctured_blocks=2)
        return a, b, c, obj.mt_get_state()

    with L.LdpcHip(LMS_DEC, H, M) as dec:
        want = run(dec)
    assert np.array_equal(want[1][0][700:1200], want[2][0])            # frames 3700..4199 either way
    for n in (2, 3, 8):
        with L.LdpcHipMulti(LMS_DEC, H, M, [0] * n) as m:
            got = run(m)
            for k in range(3):
                assert np.array_equal(got[k][0], want[k][0]) and np.array_equal(got[k][1], want[k][1]), (n, k)
            assert np.array_equal(got[3][0], want[3][0]) and got[3][1] == want[3][1]
            assert m.mt_stats()[1] == 0


@pytest.mark.gpu
@pytest.mark.parametrize("dec_id,M,snr,n,rounds,mod,punct", [
    (MS_DEC, 64, 2.0, 2, (4000, 4001, 7), 0, 0), (MS_DEC, 64, 1.5, 3, (65536, 1000, 2), 0, 0), (LMS_DEC, 126, 1.7, 5, (2500, 333), 0, 2),
    (MS_DEC, 1, 4.0, 8, (60000, 9), 0, 0), (SP_DEC, 64, 2.5, 4, (3000, 3000), 1, 1)])
def test_shard_protocol_through_the_c_abi(L, torch, dec_id, M, snr, n, rounds, mod, punct):
    """ldpc_hip_mt_shard_begin / _emit / _commit with n contexts of this process in the role of the ranks and the three exchanges
    done by hand: per round at least one rank finds the end state (all that do agree), every rank is covered, the concatenated records equal one context's
    ldpc_hip_mt_frames and so does the generator afterwards.  Rounds with fewer frames than ranks leave some ranks without frames.
    One abandoned round in between (nothing may have moved)."""
    H = relift(load_base_matrix(), M)
    key, pos = seeded_state(11)
    with L.LdpcHip(dec_id, H, M) as one:
        one.mt_set_state(key, pos)
        want = [one.mt_frames(snr, 20, f, modulation=mod, punctured_blocks=punct) for f in rounds]
        want_state = one.mt_get_state()
    ranks = [L.LdpcHip(dec_id, H, M) for _ in range(n)]
    try:
        for d in ranks:
            d.mt_set_state(key, pos)
        for k, frames in enumerate(rounds):
            if k == 1:   # a round begun and given up: state and frame index as before
                for r, d in enumerate(ranks):
                    d.mt_shard_begin(snr, frames, r, n, modulation=mod, punctured_blocks=punct)
                    d.mt_shard_abandon()
            counts = [d.mt_shard_begin(snr, frames, r, n, modulation=mod, punctured_blocks=punct) for r, d in enumerate(ranks)]
            outs = [d.mt_shard_emit(counts) for d in ranks]
            assert any(o[0] for o in outs) and all(o[1] for o in outs), [(o[0], o[1], o[3]) for o in outs]
            owner = [o[0] for o in outs].index(True)   # windows overlap in a short round: several ranks may hold the last item
            assert all(np.array_equal(o[2], outs[owner][2]) for o in outs if o[0])
            fdone = outs[owner][3]
            assert all(o[3] == fdone for o in outs) and 0 < fdone <= frames
            info, its = [], []
            for r, d in enumerate(ranks):
                lo, hi = frames * r // n, frames * (r + 1) // n
                a, b = d.mt_shard_commit(outs[owner][2], fdone, 20, max(min(hi, fdone) - lo, 0))
                info.append(a)
                its.append(b)
            info, its = np.concatenate(info), np.concatenate(its)
            if fdone < frames:   # a short round: the remainder the plain way, on every rank
                rest = [d.mt_frames(snr, 20, frames - fdone, modulation=mod, punctured_blocks=punct) for d in ranks]
                info, its = np.concatenate([info, rest[0][0]]), np.concatenate([its, rest[0][1]])
            assert np.array_equal(info, want[k][0]) and np.array_equal(its, want[k][1]), (k, frames)
        for d in ranks:
            st = d.mt_get_state()
            assert np.array_equal(st[0], want_state[0]) and st[1] == want_state[1]
            assert d.mt_frame_index() == sum(rounds)
    finally:
        for d in ranks:
            d.close()


_NCCL_ONE_RANK_WORKER = r"""
import os, sys, json
sys.path.insert(0, {root!r}); sys.path.insert(0, os.path.join({root!r}, "tests"))
import numpy as np, torch, torch.distributed as dist
import ldpc_lib_amd
from ldpc_lib_amd import host
from ldpc_testlib import load_base_matrix, relift
from test_mt_replay import _EXACT_CASES, raw_after
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
out = []
for dec_id, M, snr, maxit, n_fe, n_exp, ref, seed in _EXACT_CASES[:2]:
    H = relift(load_base_matrix(), M)
    src = host.MtFrameSource(H, M, dec_id, maxit, snr, 0, 0, seed, 0, 0.8)
    src.share_single = True
    ber, fer, st = ldpc_lib_amd.bp_simulation(H, M, maxit, n_fe, n_exp, snr, ref, decoder_type=dec_id, batch=1500, return_state=True, source=src)
    src.close()
    out.append([st["nse"], st["nde"], st["nue"], st["experiment"], st["sum_abs_iters"], ber.hex(), fer.hex(), int(raw_after(*st["generator"], 0, 1)[0]),
                st["tape_shared_rounds"], st["tape_fallback_rounds"]])
print("RESULT", json.dumps(out))
dist.destroy_process_group()
"""


@pytest.mark.gpu
def test_rank_protocol_over_rccl_with_one_rank(L, torch):
    """the exchanges of the rank-sharded exact replay (all-gather of counts, all-gather of flags, broadcast of the state, all-gather of
    the records) through torch.distributed's nccl backend = RCCL on DEVICE tensors, in a job of one rank -- what a one-GPU box can run
    of the path the N > 1 nccl job takes; results against the oracle's sequential loop."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict({k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}, MASTER_ADDR="127.0.0.1",
               MASTER_PORT=str(29400 + os.getpid() % 500), HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run([sys.executable, "-c", _NCCL_ONE_RANK_WORKER.format(root=root)], env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, (p.stdout[-2000:], p.stderr[-3000:])
    got = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("RESULT")][0].split(" ", 1)[1])
    for row, (dec_id, M, snr, maxit, n_fe, n_exp, ref, seed) in zip(got, _EXACT_CASES[:2]):
        res = _oracle_sim(relift(load_base_matrix(), M), M, dec_id, snr, maxit, n_fe, n_exp, ref, seed)
        assert row[:5] == [res.nse, res.nde, res.nue, res.experiment, res.sum_abs_iter]
        assert row[5] == float(res.ber).hex() and row[6] == float(res.fer).hex() and row[7] == res.rng_next
        assert row[8] > 0 and row[9] == 0, row
