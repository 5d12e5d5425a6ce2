"""GPU: the transmit / receive chain around the decoders and the multi-device layer behind the C-ABI.

  * mapper against the compiled reference's vectors (tests/golden/qam_frontend.npz q*_bits -> q*_sym);
  * codeword -> interleaver -> mapper -> AWGN -> demapper -> interleaver -> puncturing against a CPU twin (numpy Philox +
    the oracle's mapper / demapper + upstream's interleaver maps);
  * MS / LMS decisions mirror bit for bit under a codeword's sign flips; error counting against the sent word;
  * random codewords through QAM-16 + interleaver mode 3 + min-sum: FER within Monte-Carlo spread of the all-zero run;
  * ldpc_hip_open_multi: n = 1, 2, 8 logical shards give identical counters and ordered records; the RCCL leg on one rank.
"""
import ctypes as C
import os

import numpy as np
import pytest

from ldpc_testlib import (GOLDEN_DIR, LMS_DEC, MS_DEC, SP_DEC, Oracle, _as_double_p, load_base_matrix, oracle_lib, pack_bits,
                          philox_gauss_pairs, relift, unpack_bits)

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def L():
    import ldpc_lib_amd
    return ldpc_lib_amd


@pytest.fixture(scope="module")
def torch():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    return torch


def _random_codewords(L, H, M, count, seed):
    from ldpc_lib_amd.binding import encode
    rng = np.random.RandomState(seed)
    rh, nh = H.shape
    return np.stack([encode(H, M, rng.randint(0, 2, size=(nh - rh) * M).astype(np.uint8)) for _ in range(count)])


def test_mapper_equals_the_compiled_reference(L, torch):
    """ldpc_hip_qam_modulate_dev == QAM_modulator() on the vectors the compiled upstream code produced."""
    from ldpc_lib_amd.binding import qam_modulate
    g = np.load(os.path.join(GOLDEN_DIR, "qam_frontend.npz"))
    for Q, m in ((4, 2), (16, 4), (64, 6), (256, 8)):
        bits = torch.from_numpy(g[f"q{Q}_bits"].astype(np.uint8).reshape(-1, m)).cuda()
        sym = qam_modulate(bits, Q).cpu().numpy().ravel()
        assert np.array_equal(sym, g[f"q{Q}_sym"]), Q
        # and the oracle's restatement agrees (it is what the chain twin below uses)
        want = np.zeros_like(g[f"q{Q}_sym"])
        b = np.ascontiguousarray(g[f"q{Q}_bits"])
        oracle_lib().orc_qam_modulate(Q, _as_double_p(b), len(b), _as_double_p(want))
        assert np.array_equal(want, g[f"q{Q}_sym"])


def _chain_twin(H, M, cws, first, B, seed, snr, mod, punct, perm, punct_val):
    """CPU twin of ldpc_hip_channel_llr_dev: bp_simulation.cpp:566-577,596-710 with the device's Philox noise."""
    from ldpc_lib_amd.binding import build_interleaver
    rh, nh = H.shape
    N = nh * M
    rate = (nh - rh) / (nh - punct)
    halfmlog = 1 if mod <= 1 else mod
    m = 2 if mod <= 1 else 2 * mod
    if perm is not None:
        direct, inverse = build_interleaver(H, M, perm[0], halfmlog, perm[1], perm[2])
    else:
        direct = inverse = np.arange(N)
    units = (N + m - 1) // m
    out = np.zeros((B, N))
    Q = 1 << (2 * mod) if mod >= 1 else 1
    if mod == 0:
        sigma = np.sqrt(10.0 ** (-snr / 10.0) / 2 / rate)
    else:
        sigma = np.sqrt(10.0 ** (-snr / 10.0) / (2 * rate * halfmlog * 2) * (2.0 * (Q - 1.0) / 3.0))
    g = philox_gauss_pairs(seed, first + np.arange(B), units, tag=0 if mod <= 1 else 1)
    for b in range(B):
        cw = cws[(first + b) % len(cws)] if cws is not None else np.zeros(N, dtype=np.uint8)
        tx = np.zeros(units * m)
        tx[:N] = cw[direct]                                            # Permutation(.., 0, ..) :570, zero padding :575
        if mod <= 1:
            buf = -2.0 * (sigma * g[b, :N] + 2.0 * tx[:N] - 1.0) / (sigma * sigma)
        else:
            x = np.zeros(2 * units)
            oracle_lib().orc_qam_modulate(Q, _as_double_p(np.ascontiguousarray(tx)), len(tx), _as_double_p(x))
            x = np.ascontiguousarray(x + sigma * g[b])
            dem = np.zeros(units * m)
            oracle_lib().orc_qam_demodulate(Q, 26.0, float(sigma), _as_double_p(x), units, _as_double_p(dem), 0)
            buf = -dem[:N]
        y = buf[inverse]                                               # :684
        if punct:
            y[N - M * punct:] = punct_val                              # :697-710
        out[b] = y
    return out


@pytest.mark.parametrize("mod,snr,punct,perm", [
    (0, 1.5, 0, None), (0, 1.5, 2, (1, 128, 1)), (1, 2.0, 0, (3, 64, 1)),
    (2, 6.0, 0, (3, 64, 1)), (2, 6.0, 3, None), (3, 10.0, 1, (2, 128, 1)), (4, 14.0, 0, (4, 128, 8)), (4, 14.0, 2, (1, 128, 1)),
])
def test_channel_chain_equals_its_cpu_twin(L, torch, mod, snr, punct, perm):
    H = relift(load_base_matrix(), 64)
    cws = _random_codewords(L, H, 64, 3, seed=11 + mod)
    B, first, seed = 7, 4_000_000_123, (3 << 32) | 19
    with L.LdpcHip(MS_DEC, H, 64) as dec:
        dec.set_codewords(cws)
        if perm is not None:
            dec.set_interleaver(*perm)
        llr = dec.channel_llr(snr, seed, first, B, modulation=mod, punctured_blocks=punct).cpu().numpy()
        ref = _chain_twin(H, 64, cws, first, B, seed, snr, mod, punct, perm, 0.5)
        np.testing.assert_allclose(llr, ref, rtol=1e-9, atol=1e-9)
        # most LLR signs agree with the transmitted word (it really was sent)
        sent = np.stack([cws[(first + b) % 3] for b in range(B)])
        keep = slice(0, 2048 - 64 * punct)
        assert ((llr[:, keep] < 0) == (sent[:, keep] != 0)).mean() > 0.8
        # back to upstream's wiring: the all-zero codeword, no interleaver
        dec.set_codewords(None)
        dec.set_interleaver(0)
        llr0 = dec.channel_llr(snr, seed, first, B, modulation=mod, punctured_blocks=punct).cpu().numpy()
        np.testing.assert_allclose(llr0, _chain_twin(H, 64, None, first, B, seed, snr, mod, punct, None, 0.5), rtol=1e-9, atol=1e-9)
    with L.LdpcHip(SP_DEC, H, 64) as dec:   # probability-type decoders: punctured value 0 (sic, :700)
        if punct:
            llr = dec.channel_llr(snr, seed, first, 2, modulation=mod, punctured_blocks=punct).cpu().numpy()
            assert (llr[:, 2048 - 64 * punct:] == 0.0).all()


def test_decisions_mirror_under_the_codeword_and_are_counted_against_it(L, torch):
    """BPSK: flipping the sign of the LLRs at the codeword's ones turns the received frame into an all-zero-codeword frame with the
    mirrored noise; min-sum and layered min-sum are odd-symmetric in that flip, so decword(sent) = decword(mirrored) xor codeword,
    bit for bit with equal iteration counts, and the error counters against the SENT word equal the mirrored run's counters
    against the zero word."""
    H = relift(load_base_matrix(), 64)
    cws = _random_codewords(L, H, 64, 5, seed=3)
    B, first, seed, snr = 512, 1000, 77, 1.6
    for dec_id in (MS_DEC, LMS_DEC):
        with L.LdpcHip(dec_id, H, 64) as dec:
            dec.set_codewords(cws)
            llr1 = dec.channel_llr(snr, seed, first, B)
            sent = torch.from_numpy(np.stack([cws[(first + b) % 5] for b in range(B)])).cuda()
            # the channel is llr = -2*(sigma*g + 2*bit - 1)/sigma^2 (:603): check it against the all-zero LLRs of the same noise
            dec.set_codewords(None)
            llr_zero = dec.channel_llr(snr, seed, first, B)
            sigma = np.sqrt(10.0 ** (-snr / 10.0) / 2 / 0.5)
            np.testing.assert_allclose((llr_zero - llr1).cpu().numpy(), 4.0 / sigma ** 2 * sent.cpu().numpy(), rtol=1e-12, atol=1e-12)
            mirrored = torch.where(sent != 0, -llr1, llr1).contiguous()
            h0, it0, _ = dec.decode(mirrored, 50)
            c0, info0 = dec.count_errors(h0, it0, want_frame_info=True, first_frame=first)      # against the zero word
            dec.set_codewords(cws)
            h1, it1, _ = dec.decode(llr1, 50)
            c1, info1 = dec.count_errors(h1, it1, want_frame_info=True, first_frame=first)      # against the sent words
            torch.cuda.synchronize()
            assert torch.equal(it0, it1)
            want = h0.cpu().numpy().view(np.uint32) ^ pack_bits(sent.cpu().numpy().astype(np.float64))
            assert np.array_equal(h1.cpu().numpy().view(np.uint32), want)
            assert c0.cpu().tolist() == c1.cpu().tolist() and torch.equal(info0, info1)
            assert c0[1].item() > 0                                   # there are errored frames to count at this Eb/N0
            # fused path == composition
            s1 = dec.simulate(snr, 50, seed, first, B)
            assert [s1["nse"], s1["nde"], s1["nue"], s1["frames"], s1["sum_abs_iters"]] == c1.cpu().tolist()


def test_random_codewords_through_qam16_and_block_interleaver(L, torch):
    """random codewords from ldpc_hip_encode_host through QAM-16 + interleaver mode 3 + min-sum.
    Under 16-QAM the bit channels are not symmetric (the all-zero codeword is the constellation CORNER on every symbol, which has
    the fewest neighbours), so the all-zero run is optimistic and its FER is NOT the FER of real codewords: measured here 0.029 vs
    0.040 at 5.2 dB.  What must hold: two disjoint sets of random codewords agree within Monte-Carlo spread, the all-zero FER is
    not above them, the decoder's decisions on these frames equal the CPU oracle's, and errors are counted against the sent word."""
    H = relift(load_base_matrix(), 64)
    cws_a, cws_b = _random_codewords(L, H, 64, 16, seed=5), _random_codewords(L, H, 64, 16, seed=6)
    B, snr = 40000, 5.2
    with L.LdpcHip(MS_DEC, H, 64) as dec:
        dec.set_interleaver(3, 64, 1)
        z = dec.simulate(snr, 50, seed=9, first_frame=0, B=B, modulation=2)
        dec.set_codewords(cws_a)
        ra = dec.simulate(snr, 50, seed=9, first_frame=0, B=B, modulation=2)
        # composition == fused on a slice, decisions == the CPU oracle's on the same LLRs, counted against the SENT words
        nb = 256
        llr = dec.channel_llr(snr, 9, 0, nb, modulation=2)
        hard, iters, _ = dec.decode(llr, 50)
        cnt, info = dec.count_errors(hard, iters, first_frame=0, want_frame_info=True)
        part = dec.simulate(snr, 50, seed=9, first_frame=0, B=nb, modulation=2)
        assert [part["nse"], part["nde"], part["nue"], part["frames"], part["sum_abs_iters"]] == cnt.cpu().tolist()
        d_ref, it_ref, _ = Oracle(H, 64).decode(MS_DEC, llr.cpu().numpy(), 50, 0)
        assert np.array_equal(iters.cpu().numpy(), it_ref)
        assert np.array_equal(hard.cpu().numpy().view(np.uint32), pack_bits(d_ref))
        sent = np.stack([cws_a[b % 16] for b in range(nb)])
        wrong = d_ref.astype(np.uint8) ^ sent
        inf = info.cpu().numpy()
        assert np.array_equal((inf & (1 << 30)) != 0, wrong.any(axis=1)) and np.array_equal(inf & 0xFFFFF, wrong[:, 1024:].sum(axis=1))
        dec.set_codewords(cws_b)
        rb = dec.simulate(snr, 50, seed=10, first_frame=B, B=B, modulation=2)
    fz, fa, fb = z["nde"] / B, ra["nde"] / B, rb["nde"] / B
    assert 0.005 < fa < 0.3, fa
    spread = 4.0 * np.sqrt(fa * (1 - fa) / B * 2)
    assert abs(fa - fb) < spread, (fa, fb, spread)
    assert fz < fa + spread, (fz, fa)                 # the corner constellation point is the easy one
    assert ra["nse"] / B / 1024 < 0.05                # decoded words stay close to the sent ones


def test_puncturing_applies_behind_every_modulation(L, torch):
    """ldpc_hip_simulate with QAM16+ and punctured blocks uses the punctured bitrate and sets the tail (:444,697-710)."""
    H = relift(load_base_matrix(), 64)
    with L.LdpcHip(MS_DEC, H, 64) as dec:
        a = dec.simulate(7.0, 50, seed=4, first_frame=0, B=4096, modulation=2, punctured_blocks=0)
        b = dec.simulate(7.0, 50, seed=4, first_frame=0, B=4096, modulation=2, punctured_blocks=2)
        llr = dec.channel_llr(7.0, 4, 0, 4096, modulation=2, punctured_blocks=2)
        hard, iters, _ = dec.decode(llr, 50)
        cnt, _ = dec.count_errors(hard, iters)
        assert [b["nse"], b["nde"], b["nue"], b["frames"], b["sum_abs_iters"]] == cnt.cpu().tolist()
        assert (llr[:, 2048 - 128:] == 0.5).all()
        assert b["sum_abs_iters"] != a["sum_abs_iters"]            # a different code rate and two erased blocks: not the same run


# ---- several shards behind the C-ABI ---------------------------------------------------------------------------------------
@pytest.mark.parametrize("dec_id,M,mod,perm", [(MS_DEC, 64, 0, None), (LMS_DEC, 126, 1, (3, 126, 1)), (MS_DEC, 64, 2, (1, 128, 1))])
def test_logical_shards_give_identical_results(L, torch, dec_id, M, mod, perm):
    """n = 1, 2, 8 shards mapped to device 0: same counters, same ordered per-frame records, for any batch size; equal to the
    single-context ldpc_hip_simulate."""
    H = relift(load_base_matrix(), M)
    snr, seed, first, B = (1.7 if mod < 2 else 5.5), 31, 5000, 3000
    cws = _random_codewords(L, H, M, 4, seed=8)
    with L.LdpcHip(dec_id, H, M) as dec:
        dec.set_codewords(cws)
        if perm:
            dec.set_interleaver(*perm)
        want = dec.simulate(snr, 50, seed, first, B, modulation=mod)
        llr = dec.channel_llr(snr, seed, first, B, modulation=mod)
        hard, iters, _ = dec.decode(llr, 50)
        _, info = dec.count_errors(hard, iters, want_frame_info=True, first_frame=first)
        want_info, want_it = info.cpu().numpy(), iters.cpu().numpy()
    assert want["nde"] > 0
    for n, batch in ((1, 1000), (2, 700), (8, 128), (8, 4096), (3, 333)):
        with L.LdpcHipMulti(dec_id, H, M, [0] * n) as m:
            assert m.shards == n and m.reduction == "host"
            m.set_codewords(cws)
            if perm:
                m.set_interleaver(*perm)
            got = m.simulate(snr, 50, seed, first, B, batch, modulation=mod, records=True)
            for k in want:
                assert got[k] == want[k], (n, batch, k)
            assert np.array_equal(got["frame_info"], want_info) and np.array_equal(got["iters"], want_it)
            got2 = m.simulate(snr, 50, seed, first, B, batch, modulation=mod)
            assert all(got2[k] == want[k] for k in want)


def test_counter_allreduce_through_rccl_on_one_rank(L, torch, monkeypatch):
    """The RCCL leg (dlopen, ncclCommInitAll, ncclAllReduce of the five uint64 counters on the shard's stream) with a one-rank
    communicator -- the only RCCL configuration a one-GPU box can run; n > 1 distinct devices take the same code path."""
    monkeypatch.setenv("LDPC_HIP_RCCL_SINGLE", "1")
    H = relift(load_base_matrix(), 64)
    with L.LdpcHip(MS_DEC, H, 64) as dec:
        want = dec.simulate(1.5, 50, 2, 0, 5000)
    with L.LdpcHipMulti(MS_DEC, H, 64, [0]) as m:
        assert m.reduction == "rccl"
        got = m.simulate(1.5, 50, 2, 0, 5000, 1024)
        assert got == want
        got = m.simulate(1.5, 50, 2, 0, 5000, 5000)                  # communicator reused across calls
        assert got == want


def _build_multi_driver(tmp_path):
    """tests/cpp/multi_driver.c (plain C over the C-ABI) and tests/cpp/rccl_stub.cpp (host-memory stand-in for librccl: the box has one
    GPU and real RCCL refuses one device twice), both against /opt/rocm's HIP runtime -- no torch in that process."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe, stub = str(tmp_path / "multi_driver"), str(tmp_path / "librccl_stub.so")
    hipinc = ["-D__HIP_PLATFORM_AMD__", "-I", "/opt/rocm/include"]
    subprocess.check_call(["gcc", "-O1", *hipinc, "-I", os.path.join(root, "include"), os.path.join(root, "tests", "cpp", "multi_driver.c"), "-o", exe,
                           "-L", os.path.join(root, "ldpc-lib_amd"), "-lldpc_hip", "-L", "/opt/rocm/lib", "-lamdhip64",
                           "-Wl,-rpath," + os.path.join(root, "ldpc-lib_amd"), "-Wl,-rpath,/opt/rocm/lib"])
    subprocess.check_call(["g++", "-O1", "-fPIC", "-shared", *hipinc, os.path.join(root, "tests", "cpp", "rccl_stub.cpp"), "-o", stub,
                           "-L", "/opt/rocm/lib", "-lamdhip64", "-Wl,-rpath,/opt/rocm/lib"])
    return exe, stub


def _run_multi_driver(exe, n, env_extra, timeout=240):
    import subprocess
    env = {k: v for k, v in os.environ.items() if not k.startswith("LDPC_HIP_")}
    env.update(env_extra)
    p = subprocess.run([exe, os.path.join(GOLDEN_DIR, "h16x32_m126.txt"), "64", str(n)], env=env, text=True, capture_output=True, timeout=timeout)
    return p.returncode, p.stdout.splitlines(), p.stderr


def test_communicator_path_with_several_ranks_through_a_stand_in_rccl(L, tmp_path):
    """The N > 1 communicator code of csrc/ldpc_multi.hpp -- cached ncclCommInitAll, ONE grouped all-reduce for all ranks after every
    shard has enqueued, results read back per rank and compared -- for n = 2, 3, 8 ranks on the one GPU, through a host-memory
    stand-in for librccl that reports an incomplete or ungrouped collective as an error (the real library would hang).  Counters,
    ordered records and the resident-batch entry point equal the n = 1 host-sum run; a second multi context on the same device list
    makes no second communicator."""
    exe, stub = _build_multi_driver(tmp_path)
    rc, want, err = _run_multi_driver(exe, 1, {})
    assert rc == 0, (want, err)
    assert want[0].startswith("shards 1 reduction host")
    body = lambda lines: [ln for ln in lines if ln.split()[0] in ("simulate", "frames", "decode_count", "second")]
    assert len(body(want)) == 5 and int(want[1].split()[2]) > 0                      # some frames fail at 2.0 dB
    for n in (2, 3, 8):
        rc, got, err = _run_multi_driver(exe, n, {"LDPC_HIP_RCCL_PATH": stub, "LDPC_HIP_RCCL_ALLOW_DUPLICATE": "1"})
        assert rc == 0, (n, got, err)
        assert got[0] == f"shards {n} reduction rccl comm_inits 1", got[0]
        assert body(got) == body(want), (n, got, want)
        assert got[-1] == "comm_inits 1", got                                        # the second context reused the communicators


@pytest.mark.parametrize("n,bad", [(2, 1), (8, 5), (3, 0)])
def test_a_failing_shard_fails_the_call_instead_of_hanging_its_peers(L, tmp_path, n, bad):
    """ADVICE r2 / VERDICT r2: a shard that fails before the collective used to leave its peers inside ncclAllReduce for ever.  With
    the two-phase run no rank enters the all-reduce unless every shard enqueued: every counting call returns the failing shard's
    error (the stand-in library would flag an incomplete collective, a real one would hang -> the timeout below)."""
    exe, stub = _build_multi_driver(tmp_path)
    rc, out, err = _run_multi_driver(exe, n, {"LDPC_HIP_RCCL_PATH": stub, "LDPC_HIP_RCCL_ALLOW_DUPLICATE": "1", "LDPC_HIP_TEST_FAIL_SHARD": str(bad)},
                                     timeout=120)
    assert rc == 3, (rc, out, err)
    failed = [ln for ln in out if "failed:" in ln]
    assert len(failed) == 5 and all(f"shard {bad} " in ln and "injected failure" in ln for ln in failed), out
    assert not any("invalid usage" in ln for ln in out)                              # the collective was never entered incomplete


def test_decode_host_over_shards_and_exact_harness(L, torch):
    """ldpc_hip_decode_host_multi (contiguous slices, one host thread per shard) == the single-context call; and the exact-replay
    C++ harness run with LDPC_HIP_DEVICES=0,0,0 returns the sequential harness's counters and generator state."""
    g = np.load(os.path.join(GOLDEN_DIR, "ms_m64_1p2.npz"))
    H, M, maxiter = g["H"], int(g["M"]), int(g["maxiter"])
    with L.LdpcHipMulti(MS_DEC, H, M, [0, 0, 0]) as m:
        dec, its, _ = m.decode_host(g["llr"], maxiter)
        assert np.array_equal(its, g["iters"]) and np.array_equal(pack_bits(dec), g["hard"])
    from test_gpu_parity import _compat_lib
    from ldpc_testlib import SimResult, c_int_p
    lib = _compat_lib(L)
    Hm = np.ascontiguousarray(relift(load_base_matrix(), 64), dtype=np.int32)
    out = (C.c_double * 7)()
    nxt = C.c_uint()
    os.environ["LDPC_HIP_DEVICES"] = "0,0,0"
    try:
        assert lib.ldpc_bp_simulation_exact(16, 32, Hm.ctypes.data, 64, 50, 10**9, 1500, 1.2, 0.02, MS_DEC, 0, 0, 1, 0, C.addressof(out), C.addressof(nxt)) == 0
    finally:
        del os.environ["LDPC_HIP_DEVICES"]
    res = SimResult()
    assert oracle_lib().orc_bp_simulation(16, 32, Hm.ctypes.data_as(c_int_p), 64, 50, 10**9, 1500, 1.2, 0.02, MS_DEC, 0, 0, 1, C.byref(res), None) == 0
    assert (out[2], out[3], out[4], out[5], out[6]) == (res.nse, res.nde, res.nue, res.experiment, res.sum_abs_iter)
    assert nxt.value == res.rng_next


@pytest.mark.parametrize("n_fe,n_exp,ref,devs,batch", [(10**9, 20000, 1.0, [0], 4096), (40, 10**7, 1.0, [0, 0], 1024), (10**9, 30000, 0.004, [0, 0, 0, 0], 512)])
def test_cpp_throughput_harness_equals_the_python_harness(L, torch, n_fe, n_exp, ref, devs, batch):
    """ldpc::bp_simulation_throughput_t (C++ over ldpc_hip_frames_multi) and ldpc_lib_amd.bp_simulation (Python over torch tensors)
    replay the same sequential rule over the same device noise: identical counters whatever the sharding."""
    from test_gpu_parity import _compat_lib
    lib = _compat_lib(L)
    H = np.ascontiguousarray(relift(load_base_matrix(), 64), dtype=np.int32)
    out = (C.c_double * 7)()
    d = (C.c_int * len(devs))(*devs)
    lib.ldpc_bp_simulation_throughput.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_longlong, C.c_double, C.c_double,
                                                  C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_ulonglong, C.c_void_p, C.c_int,
                                                  C.c_longlong, C.c_void_p, C.c_int, C.c_void_p]
    assert lib.ldpc_bp_simulation_throughput(16, 32, H.ctypes.data, 64, 50, n_fe, n_exp, 1.8, ref, MS_DEC, 0, 0, 128, 1, 0, 123, d, len(devs),
                                             batch, None, 0, C.addressof(out)) == 0
    ber, fer, st = L.bp_simulation(H, 64, 50, n_fe, n_exp, 1.8, ref, decoder_type=MS_DEC, seed=123, batch=3000, return_state=True)
    assert (out[2], out[3], out[4], out[5], out[6]) == (st["nse"], st["nde"], st["nue"], st["experiment"], st["sum_abs_iters"])
    assert out[0] == ber and out[1] == fer


_RANK_WORKER = r"""
import os, sys, json
sys.path.insert(0, {root!r}); sys.path.insert(0, os.path.join({root!r}, "tests"))
import torch, torch.distributed as dist
import ldpc_lib_amd
from ldpc_testlib import load_base_matrix, relift
dist.init_process_group("gloo", rank=int(os.environ["RANK"]), world_size=int(os.environ["WORLD_SIZE"]))
H = relift(load_base_matrix(), 64)
out = []
for n_fe, n_exp, ref, batch in [(10**9, 6000, 1.0, 1024), (30, 10**6, 1.0, 500), (10**9, 20000, 0.004, 2048)]:
    _, _, st = ldpc_lib_amd.bp_simulation(H, 64, 50, n_fe, n_exp, 1.8, ref, seed=123, batch=batch, return_state=True)
    out.append([st["nse"], st["nde"], st["nue"], st["experiment"], st["sum_abs_iters"]])
print("RESULT", dist.get_rank(), json.dumps(out))
dist.destroy_process_group()
"""


def test_two_gpu_ranks_with_the_real_decoder(L, torch):
    """The N > 1 path of ldpc_lib_amd.bp_simulation / bench.py with the REAL GpuFrameSource: two processes (both on this box's one
    GPU, gloo for the record exchange) must each report the single-process result."""
    import json
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    H = relift(load_base_matrix(), 64)
    want = []
    for n_fe, n_exp, ref in [(10**9, 6000, 1.0), (30, 10**6, 1.0), (10**9, 20000, 0.004)]:
        _, _, st = L.bp_simulation(H, 64, 50, n_fe, n_exp, 1.8, ref, seed=123, batch=4096, return_state=True)
        want.append([st["nse"], st["nde"], st["nue"], st["experiment"], st["sum_abs_iters"]])
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, "-c", _RANK_WORKER.format(root=root)], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=600)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    for o in outs:
        line = [ln for ln in o.splitlines() if ln.startswith("RESULT")][0]
        assert json.loads(line.split(" ", 2)[2]) == want


def test_ldpc_sim_throughput_mode_with_interleaver_qam_and_device_list(L, torch, tmp_path):
    """`ldpc_sim simulation --throughput --devices 0,0`: a jsonx scenario with 16-QAM + interleaver mode 3 (round 1 refused every
    permutation_type but 0 in throughput mode) must give the FER / BER the C-ABI gives for the same noise, on two logical shards."""
    import subprocess
    from test_gpu_parity import _compat_lib
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    _compat_lib(L)
    exe = os.path.join(root, "ldpc-lib_amd", "ldpc_sim")
    src = open(os.path.join(root, "examples", "simulation_appendix_c.jsonx")).read()
    src = src.replace("modulation_type = 0", "modulation_type = 2").replace("permutation_type = 0", "permutation_type = 3")
    src = src.replace("permutation_block = 128", "permutation_block = 64").replace("num_codewords = 2000", "num_codewords = 30000")
    scen = tmp_path / "scen.jsonx"
    scen.write_text(src)
    out = str(tmp_path / "res.jsonx")
    subprocess.check_call([exe, "simulation", str(scen), out, "--throughput", "--devices", "0,0"], stdout=subprocess.DEVNULL)

    def get(path):
        return subprocess.check_output([exe, "jsonx-get", out, path], text=True).strip()

    def numbers(path):
        return [float(x) for x in get(path).replace("array {", "").replace("}", "").split()]
    # code #3 of the example: (2048,1024) min-sum 50 it at 2.0 and 2.5 dB -> the same points through the C-ABI on one context
    assert int(get("results/3/_decoder_type")) == MS_DEC and int(get("results/3/_lifting")) == 64
    snrs, fer, ber = numbers("results/3/_SNRs"), numbers("results/3/simulation_logs/0/FER"), numbers("results/3/simulation_logs/0/BER")
    H = relift(load_base_matrix(), 64)
    with L.LdpcHip(MS_DEC, H, 64) as dec:
        dec.set_interleaver(3, 64, 1)
        for snr, f, b in zip(snrs, fer, ber):
            s = dec.simulate(snr, 50, seed=1, first_frame=0, B=30001, modulation=2)
            assert f == s["nde"] / 30001 and b == s["nse"] / 30001 / 1024, (snr, f, s)
            assert s["nde"] > 0


def test_bench_gpus_2_runs_without_a_launcher(L):
    """`python bench.py --gpus 2` exactly as a driver without torchrun would start it: the parent (no GPU state) starts the two ranks
    (both on this box's one GPU: --backend gloo, counters reduced through the host), relays rank 0's line and adds the abi_multi
    leg -- ldpc_hip_open_multi + ldpc_hip_decode_count_multi in one fresh process, two shards."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--backend", "gloo", "--steps", "3", "--warmup", "1",
                        "--frames", "8192", "--no-extras"], env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, (p.stdout[-2000:], p.stderr[-2000:])
    line = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["steps"] == 3 and line["value"] > 0 and line["scaling"] == "weak"
    assert line["config"]["global_frames_per_step"] == 2 * 8192
    abi = line["abi_multi"]
    assert "error" not in abi, abi
    assert abi["n_gpus"] == 2 and abi["reduction"] == "host" and abi["value"] > 0 and abi["frames_per_gpu_per_step"] == 8192
    assert abs(abi["fer"] - line["config"]["fer"]) < 0.01 and abi["mean_iters_per_frame"] > 49.9   # the same workload: 0 dB, all iterations


@pytest.mark.gpu
def test_bench_exact_ranks_leg_is_independent_of_the_rank_count(L):
    """bench.py's exact_replay_ranks leg (host.bp_simulation(exact_seed=1) as a one-process-per-GPU job, the generator's tape shared
    out over the ranks): three ranks on this box's one GPU (gloo) must report the counters and the generator end state of one rank."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    # as if called from a rank that torch.distributed.run started: the leg's own job must not inherit that launcher's rendezvous
    env.update(TORCHELASTIC_USE_AGENT_STORE="True", TORCHELASTIC_RUN_ID="outer", GROUP_RANK="0")
    lines = {}
    for n in (1, 3):
        p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--leg", "exact_ranks", "--gpus", str(n), "--steps", "4"], env=env,
                           capture_output=True, text=True, timeout=600)
        assert p.returncode == 0, (p.stdout[-2000:], p.stderr[-2000:])
        lines[n] = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])
        assert "error" not in lines[n], lines[n]
    same = ("frames", "errored_frames", "bit_errors", "sum_iterations", "generator_state_crc32", "generator_next_index")
    assert lines[1]["frames"] == 4 * 65536 and [lines[1][k] for k in same] == [lines[3][k] for k in same]
    assert lines[3]["tape_shared_rounds"] >= 4 and lines[3]["tape_fallback_rounds"] == 0 and lines[1]["tape_shared_rounds"] == 0


def test_multi_gpu_c_example_runs(L, tmp_path):
    """examples/simulate_multi.c end to end (every GPU of the box, RCCL when there is more than one... one here): random codewords,
    16-QAM, block interleaver; FER in the range the Python route measures for the same point."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "simulate_multi")
    subprocess.check_call(["gcc", "-O1", "-I", os.path.join(root, "include"), os.path.join(root, "examples", "simulate_multi.c"), "-o", exe,
                           "-L", os.path.join(root, "ldpc-lib_amd"), "-lldpc_hip", "-Wl,-rpath," + os.path.join(root, "ldpc-lib_amd")])
    out = subprocess.check_output([exe, os.path.join(GOLDEN_DIR, "h16x32_m126.txt"), "64", "3", "50", "5.2", "60000"], text=True)
    row = [ln for ln in out.splitlines() if ln.startswith("Eb/N0")][0].split()
    fer = float(row[row.index("FER") + 1])
    assert int(row[row.index("frames") + 1]) == 60000 and 0.02 < fer < 0.07, out


# ---- the encoder on the device (csrc/ldpc_encode.hpp) -----------------------------------------------------------------------------
@pytest.mark.parametrize("M", [1, 7, 64, 126, 512])
def test_device_encoder_equals_the_host_encoder(L, torch, M):
    """ldpc_hip_encode_dev == ldpc_hip_encode_host (upstream's qc_encode restated), bit for bit, and every word is a codeword"""
    from ldpc_lib_amd.binding import encode
    H = relift(load_base_matrix(), M)
    rh, nh = H.shape
    rng = np.random.RandomState(1000 + M)
    info = rng.randint(0, 2, size=(37, (nh - rh) * M)).astype(np.uint8)
    with L.LdpcHip(MS_DEC, H, M) as dec:
        got = dec.encode_dev(torch.from_numpy(info).cuda()).cpu().numpy()
    want = np.stack([encode(H, M, row) for row in info])
    assert np.array_equal(got, want)
    assert np.array_equal(got[:, rh * M:], info)                      # systematic
    for w in got[:5]:                                                  # H c = 0 (bp_simulation.cpp:87-116)
        for i in range(rh):
            s = np.zeros(M, dtype=np.uint8)
            for j in range(nh):
                if H[i, j] >= 0:
                    s ^= np.roll(w[j * M:(j + 1) * M], -int(H[i, j]) % M)
            assert not s.any()


@pytest.mark.parametrize("M,blocks", [(64, (4, 4)), (7, (3, 5, 4)), (126, (5, 3)), (2, (4, 4, 4))])
def test_device_encoder_on_several_dual_diagonal_blocks(L, torch, M, blocks):
    """bp_simulation.cpp:142-191: a parity part made of several bidiagonal blocks is encoded from the last block to the first, each
    against the information part and the parity of the blocks behind it.  Round 2's device encoder refused such matrices; now
    ldpc_hip_encode_dev == ldpc_hip_encode_host on them, every word is a codeword, and a table of random codewords made on the
    device decodes to itself."""
    from ldpc_lib_amd.binding import encode
    from ldpc_testlib import multi_block_code, syndrome_np
    rng = np.random.RandomState(77 + M)
    H = multi_block_code(rng, M, blocks)
    b = sum(blocks)
    info = rng.randint(0, 2, size=(29, (H.shape[1] - b) * M)).astype(np.uint8)
    want = np.stack([encode(H, M, row) for row in info])
    assert not syndrome_np(H, M, want).any()
    with L.LdpcHip(MS_DEC, H, M) as dec:
        got = dec.encode_dev(torch.from_numpy(info).cuda()).cpu().numpy()
        assert np.array_equal(got, want)
        dec.set_random_codewords(5, 40)
        s = dec.simulate(12.0, 30, seed=3, first_frame=0, B=400)
        assert s["frames"] == 400 and s["nde"] == 0


def test_device_encoder_refuses_what_it_does_not_cover(L, torch):
    H = relift(load_base_matrix(), 64).copy()
    H[3, 2] = -1                      # breaks the double diagonal: not encodable at all
    with L.LdpcHip(MS_DEC, H, 64) as dec:
        with pytest.raises(L.LdpcHipError):
            dec.set_random_codewords(1, 16)


@pytest.mark.parametrize("mod,perm,snr", [(0, 0, 9.0), (2, 3, 14.0), (1, 1, 9.0)])
def test_random_codeword_table_from_the_device(L, torch, mod, perm, snr):
    """ldpc_hip_set_random_codewords: table, transmit order and packed reference words are all made on the device.  At a high
    Eb/N0 every frame must decode to ITS codeword (zero errors against the sent words -- any inconsistency between the three would
    show as errors), the codewords differ from frame to frame, and the same seed gives the same table on a second context."""
    M = 64
    H = relift(load_base_matrix(), M)
    with L.LdpcHip(MS_DEC, H, M) as dec:
        if perm:
            dec.set_interleaver(perm, 64, 1)
        dec.set_random_codewords(77, 300)
        s = dec.simulate(snr, 50, seed=5, first_frame=0, B=1500, modulation=mod)
        assert s["frames"] == 1500 and s["nde"] == 0 and s["nse"] == 0
        llr = dec.awgn_llr(snr, 5, 0, 600, modulation=mod)
        hard, iters, _ = dec.decode(llr, 50)
        bits = unpack_bits(hard.cpu().numpy().view(np.uint32), dec.N)
        assert np.array_equal(bits[:300], bits[300:600])               # frame f carries codeword f % 300
        assert len({row.tobytes() for row in bits[:300]}) == 300 and bits[:300].mean() > 0.4
        with L.LdpcHip(MS_DEC, H, M) as dec2:
            dec2.set_random_codewords(77, 300)
            hard2, _, _ = dec2.decode(dec2.awgn_llr(snr, 5, 0, 300, modulation=0), 50)
            assert np.array_equal(unpack_bits(hard2.cpu().numpy().view(np.uint32), dec.N), bits[:300])
        dec.set_random_codewords(0, 0)                                  # back to upstream's all-zero codeword
        s = dec.simulate(snr, 50, seed=5, first_frame=0, B=200, modulation=mod)
        assert s["nde"] == 0


def test_ldpc_sim_with_codewords_encoded_on_the_device(L, torch, tmp_path):
    """`ldpc_sim simulation --throughput --random-codewords 128`: errors are counted against 128 different sent words made on the
    device; for BPSK the code's symmetry makes the FER of the min-sum points statistically the all-zero run's (not identical: the same
    noise sample hits a different transmitted sign)."""
    import subprocess
    from test_gpu_parity import _compat_lib
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    _compat_lib(L)
    exe = os.path.join(root, "ldpc-lib_amd", "ldpc_sim")
    src = open(os.path.join(root, "examples", "simulation_appendix_c.jsonx")).read().replace("num_codewords = 2000", "num_codewords = 40000")
    scen = tmp_path / "scen.jsonx"
    scen.write_text(src)
    fers = []
    for extra in ([], ["--random-codewords", "128"]):
        out = str(tmp_path / ("res%d.jsonx" % len(fers)))
        subprocess.check_call([exe, "simulation", str(scen), out, "--throughput"] + extra, stdout=subprocess.DEVNULL)
        get = lambda path, o=out: subprocess.check_output([exe, "jsonx-get", o, path], text=True).strip()
        assert int(get("results/3/_decoder_type")) == MS_DEC
        fers.append([float(x) for x in get("results/3/simulation_logs/0/FER").replace("array {", "").replace("}", "").split()])
    assert fers[0] != fers[1]                                   # other words were sent
    for a, b in zip(*fers):                                      # 40001 frames per point: sd of the difference ~ 0.0014 at FER 0.04
        assert a > 0 and abs(a - b) < 0.007, fers
