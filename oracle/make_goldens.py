#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ from the COMPILED UPSTREAM REFERENCE (oracle/_ref).

Run in the build container only (needs /root/reference to have built oracle/_ref via `make -C oracle ref`):

    python3 oracle/make_goldens.py

The reference has no golden vectors or known-answer tests of its own for the decoder path,
so these fixtures -- inputs and the reference's outputs on them -- are what pins parity.  Fixtures are data
only: LLR inputs (float64, in the reference's own mt19937/normal_distribution draw order, seed 1), the
decoder's signed iteration count, hard decisions (packed), and a-posteriori soft values for a few frames.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from ldpc_testlib import (ASP_DEC, BP_DEC, GOLDEN_DIR, IMS_DEC, LMS_DEC, MS_DEC, SP_DEC, TASP_DEC, Reference, awgn_llr, load_base_matrix,  # noqa: E402
                          oracle_lib, pack_bits, ref_lib, relift, _as_double_p)

SETS = [
    # name,            dec,     M,   snr, frames, maxiter, soft_frames
    ("ms_m64_2p0",     MS_DEC,  64,  2.0, 24, 50, 4),
    ("ms_m64_1p2",     MS_DEC,  64,  1.2, 24, 50, 4),   # low SNR: non-converging frames, where rounding order matters
    ("ms_m64_0p0",     MS_DEC,  64,  0.0, 6,  50, 2),   # worst case: every frame runs all 50 iterations
    ("ms_m1_4p0",      MS_DEC,  1,   4.0, 128, 20, 16),  # BASELINE config #1 (32,16)
    ("ms_m126_1p7",    MS_DEC,  126, 1.7, 8,  50, 2),   # the lifting of the shipped `search` scenarios
    ("ms_m32_3p0",     MS_DEC,  32,  3.0, 16, 50, 2),
    ("ms_m512_1p6",    MS_DEC,  512, 1.6, 3,  50, 1),
    ("lms_m64_1p6",    LMS_DEC, 64,  1.6, 24, 50, 4),
    ("lms_m64_0p8",    LMS_DEC, 64,  0.8, 12, 50, 2),
    ("lms_m512_1p6",   LMS_DEC, 512, 1.6, 4,  50, 1),   # BASELINE config #4 (16384,8192)
    ("lms_m512_1p0",   LMS_DEC, 512, 1.0, 2,  50, 1),
    ("lms_m1_4p0",     LMS_DEC, 1,   4.0, 64, 20, 8),
    ("sp_m64_2p0",     SP_DEC,  64,  2.0, 24, 50, 4),   # BASELINE config #3
    ("sp_m64_1p2",     SP_DEC,  64,  1.2, 16, 50, 4),
    ("sp_m1_4p0",      SP_DEC,  1,   4.0, 64, 20, 8),
    ("ims_m64_2p0",    IMS_DEC, 64,  2.0, 16, 50, 4),
    ("tasp_m126_1p7",  TASP_DEC, 126, 1.7, 12, 15, 2),  # the shipped search scenario: decoder_type 7, M 126, 15 iterations, 1.7 dB
    ("tasp_m64_1p7",   TASP_DEC, 64,  1.7, 24, 15, 4),
    ("tasp_m64_1p0",   TASP_DEC, 64,  1.0, 12, 50, 2),
    ("tasp_m1_4p0",    TASP_DEC, 1,   4.0, 64, 20, 8),
    ("asp_m64_2p0",    ASP_DEC, 64,  2.0, 24, 50, 4),   # probability-domain flooding sum-product (decoder 2)
    ("asp_m64_1p2",    ASP_DEC, 64,  1.2, 12, 50, 2),
    ("asp_m128_1p7",   ASP_DEC, 128, 1.7, 8,  30, 2),
    ("bp_m64_2p0",     BP_DEC,  64,  2.0, 24, 50, 4),   # Gallager BP in the log domain (decoder 0)
    ("bp_m64_1p0_stale", BP_DEC, 64, 1.0, 16, 50, 2),   # failed frames followed by frames that are codewords at the input:
                                                        # upstream's uncleared syndrome array makes those return 1, not 0 (Q8)
    ("bp_m128_1p7",    BP_DEC,  128, 1.7, 8,  30, 2),
]
# sets on other protographs than the example code: (name, decoder, code factory, M, snr, frames, maxiter, soft_frames)
def _cycle_code(M):
    from ldpc_testlib import cycle_code
    return cycle_code(np.random.RandomState(1), 4, 8, M)


EXTRA_SETS = [
    # every block column holds exactly two circulants: sum_prod_gf2_decod_qc_lm's own branch (asp_all_cw_2, decoders.cpp:1027-1044, :2431-2480)
    ("asp_cw2_m64_2p0", ASP_DEC, _cycle_code, 64, 2.0, 24, 40, 4),
]
ONLY = set(sys.argv[1:])  # optional: regenerate just the named sets


def main():
    if ref_lib() is None:
        sys.exit("oracle/_ref/libldpc_ref.so missing: run `make -C oracle ref` where /root/reference exists")
    H0 = load_base_matrix()
    os.makedirs(GOLDEN_DIR, exist_ok=True)
    for name, dec, M, snr, frames, maxiter, soft_frames in SETS:
        if ONLY and name not in ONLY:
            continue
        H = relift(H0, M)
        llr = awgn_llr(H, M, snr, 1, frames)
        if name.endswith("_stale"):
            clean = awgn_llr(H, M, 15.0, 2, frames)   # no channel errors: codewords at the decoder input
            for f in (0, 2, 6, 9, 10):               # 2, 6, 9 follow failed frames; 0 and 10 do not
                llr[f] = clean[f]
        ref = Reference(dec, H, M)
        dec0, it0, after0 = ref.decode(dec, llr, maxiter, 0)
        dec1, it1, _ = ref.decode(dec, llr[:soft_frames], maxiter, 1)
        assert np.array_equal(it0[:soft_frames], it1)
        np.savez_compressed(
            os.path.join(GOLDEN_DIR, name + ".npz"),
            H=H.astype(np.int16), M=np.int32(M), dec_id=np.int32(dec), snr=np.float64(snr), maxiter=np.int32(maxiter),
            llr=llr, iters=it0.astype(np.int32), hard=pack_bits(dec0), soft=dec1,
        )
        print(f"{name}: frames={frames} iters={it0.tolist()[:12]}... fail={(it0 < 0).sum()} errbits={int((dec0 != 0).sum())}")
        ref.close()

    for name, dec, factory, M, snr, frames, maxiter, soft_frames in EXTRA_SETS:
        if ONLY and name not in ONLY:
            continue
        H = factory(M)
        llr = awgn_llr(H, M, snr, 1, frames)
        ref = Reference(dec, H, M)
        dec0, it0, after0 = ref.decode(dec, llr, maxiter, 0)
        dec1, it1, _ = ref.decode(dec, llr[:soft_frames], maxiter, 1)
        assert np.array_equal(it0[:soft_frames], it1)
        np.savez_compressed(
            os.path.join(GOLDEN_DIR, name + ".npz"),
            H=H.astype(np.int16), M=np.int32(M), dec_id=np.int32(dec), snr=np.float64(snr), maxiter=np.int32(maxiter),
            llr=llr, iters=it0.astype(np.int32), hard=pack_bits(dec0), soft=dec1,
        )
        print(f"{name}: frames={frames} iters={it0.tolist()[:12]}... fail={(it0 < 0).sum()} errbits={int((dec0 != 0).sum())}")
        ref.close()

    if ONLY and "qam_frontend" not in ONLY:
        return
    # QAM front end (QAM_modulator.cpp / QAM_demodulator.cpp), function level.
    rlib = ref_lib()
    rng = np.random.RandomState(7)
    out = {}
    for Q, m in ((4, 2), (16, 4), (64, 6), (256, 8)):
        nbits = 64 * m
        bits = rng.randint(0, 2, nbits).astype(np.float64)
        sym = np.zeros(2 * (nbits // m), dtype=np.float64)
        ns = rlib.ref_qam_modulate(Q, _as_double_p(bits), nbits, _as_double_p(sym))
        assert ns == nbits // m
        out[f"q{Q}_bits"] = bits
        out[f"q{Q}_sym"] = sym
        for out_type in (0, 1):
            for sigma in (0.35, 0.8, 2.5):
                x = sym + sigma * rng.randn(sym.size)
                x[:4] = sym[:4] + np.array([40.0, -40.0, 17.0, -17.0])  # force the cut-off branches (p0==0 / p1==0 / 0/0)
                llr = np.zeros(ns * m, dtype=np.float64)
                rlib.ref_qam_demodulate(Q, 26.0, sigma, _as_double_p(x), ns, _as_double_p(llr), out_type)
                key = f"q{Q}_t{out_type}_s{str(sigma).replace('.', 'p')}"
                out[key + "_x"] = x
                out[key + "_llr"] = llr
    np.savez_compressed(os.path.join(GOLDEN_DIR, "qam_frontend.npz"), **out)
    print("qam_frontend:", sorted(out.keys()))


if __name__ == "__main__":
    main()
