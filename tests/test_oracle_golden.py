"""CPU: the C restatement (oracle/) reproduces every golden vector that oracle/make_goldens.py produced from the
compiled upstream reference -- hard bits, signed iteration counts and a-posteriori soft values, bit for bit."""
import glob
import os

import numpy as np
import pytest

from ldpc_testlib import (BP_DEC, GOLDEN_DIR, SP_DEC, TASP_DEC, Oracle, awgn_llr, load_base_matrix, oracle_lib, pack_bits, relift,
                          _as_double_p)

DECODER_SETS = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN_DIR, "*.npz"))
                      if not os.path.basename(p).startswith(("qam", "interleavers")))


def test_golden_sets_present():
    assert len(DECODER_SETS) >= 12
    assert os.path.exists(os.path.join(GOLDEN_DIR, "qam_frontend.npz"))


@pytest.mark.parametrize("name", DECODER_SETS)
def test_oracle_matches_reference_golden(name):
    g = np.load(os.path.join(GOLDEN_DIR, name + ".npz"))
    H, M, dec_id, maxiter = g["H"], int(g["M"]), int(g["dec_id"]), int(g["maxiter"])
    o = Oracle(H, M)
    dec, its, _ = o.decode(dec_id, g["llr"], maxiter, 0)
    assert np.array_equal(its, g["iters"])
    assert np.array_equal(pack_bits(dec), g["hard"])
    nsoft = g["soft"].shape[0]
    soft, its1, _ = o.decode(dec_id, g["llr"][:nsoft], maxiter, 1)
    assert np.array_equal(its1, g["iters"][:nsoft])
    if dec_id == TASP_DEC:
        # upstream ignores `decision` for this decoder (decoders.cpp:2737-2738): the golden "soft" output is the hard
        # decision again; the oracle's decision=1 mode returns the a-posteriori probabilities it thresholds
        assert np.array_equal((soft > 0.5).astype(np.float64), g["soft"])
        return
    # same libm on both sides in this container, so even sum-product is bit-identical here
    assert np.array_equal(soft, g["soft"], equal_nan=True)


def test_qam_frontend_matches_reference_golden():
    g = np.load(os.path.join(GOLDEN_DIR, "qam_frontend.npz"))
    lib = oracle_lib()
    for Q, m in ((4, 2), (16, 4), (64, 6), (256, 8)):
        bits = np.ascontiguousarray(g[f"q{Q}_bits"])
        sym = np.zeros(2 * (bits.size // m))
        ns = lib.orc_qam_modulate(Q, _as_double_p(bits), bits.size, _as_double_p(sym))
        assert ns == bits.size // m
        assert np.array_equal(sym, g[f"q{Q}_sym"])
        for out_type in (0, 1):
            for s in ("0p35", "0p8", "2p5"):
                key = f"q{Q}_t{out_type}_s{s}"
                x = np.ascontiguousarray(g[key + "_x"])
                llr = np.zeros(ns * m)
                lib.orc_qam_demodulate(Q, 26.0, float(s.replace("p", ".")), _as_double_p(x), ns, _as_double_p(llr), out_type)
                assert np.array_equal(llr, g[key + "_llr"], equal_nan=True), key


def test_bp_outputs_survive_one_ulp_perturbations(tmp_path):
    """The device computes BP's exp()/log() with ocml, the reference with glibc (both < 1 ulp).  Rebuild the restatement with
    every exp/log result moved by -1/0/+1 ulp at random (oracle/bp_perturb.h): iteration counts and hard decisions of
    converging AND failing frames stay the same, which is what lets the GPU tests demand equality for them."""
    import ctypes as C
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    so = str(tmp_path / "liboracle_pert.so")
    subprocess.check_call(["gcc", "-O2", "-ffp-contract=off", "-fPIC", "-shared", "-include", os.path.join(root, "oracle", "bp_perturb.h"),
                           "-I", os.path.join(root, "oracle"), os.path.join(root, "oracle", "ldpc_oracle.c"), "-o", so, "-lm"])
    pl = C.CDLL(so)
    pl.orc_open.restype = C.c_void_p
    pl.orc_open.argtypes = [C.c_int, C.c_int, C.c_int, C.c_void_p]
    pl.orc_bp.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int]
    H = relift(load_base_matrix(), 64)
    Hs = np.ascontiguousarray(H, dtype=np.int16)
    for snr, frames in ((0.8, 60), (1.4, 120), (2.0, 60)):
        llr = awgn_llr(H, 64, snr, 7, frames)
        d1, i1, _ = Oracle(H, 64).decode(BP_DEC, llr, 50, 0)
        h = pl.orc_open(16, 32, 64, Hs.ctypes.data)
        i2 = np.zeros(frames, dtype=np.int32)
        d2 = np.zeros_like(llr)
        for f in range(frames):
            y = llr[f].copy()
            i2[f] = pl.orc_bp(h, y.ctypes.data, d2[f].ctypes.data, 50, 0)
        assert np.array_equal(i1, i2) and np.array_equal(d1, d2)
        assert snr > 1.0 or (i1 < 0).sum() >= frames // 3   # the low point holds many failing frames


@pytest.mark.parametrize("dec_id,M,maxit,n_exp,snr,nde,experiment", [
    (3, 64, 50, 4000, 2.0, 170, 4001),     # cfg2, the headline configuration (FER 0.0425)
    (3, 1, 20, 2000, 4.0, 112, 2001),      # cfg1
    (1, 64, 50, 2000, 2.0, 13, 2001),      # cfg3 sum-product
])
def test_harness_restatement_reproduces_the_surveys_upstream_counts(dec_id, M, maxit, n_exp, snr, nde, experiment):
    """errored-frame counts the survey measured with the upstream binary (seed 1, all-zero codeword,
    BPSK).  The sequential harness restatement must return exactly these; the GPU harness is then compared with it."""
    import ctypes as C
    from ldpc_testlib import SimResult, c_int_p
    H = np.ascontiguousarray(relift(load_base_matrix(), M), dtype=np.int32)
    res = SimResult()
    assert oracle_lib().orc_bp_simulation(16, 32, H.ctypes.data_as(c_int_p), M, maxit, 10**9, n_exp, snr, 1.0, dec_id, 0, 0, 1,
                                          C.byref(res), None) == 0
    assert (res.nde, res.experiment) == (nde, experiment)
    assert res.fer == nde / experiment
