"""CPU, build container only: the C restatement against the compiled upstream reference (oracle/_ref) on fresh
random batches beyond the committed goldens.  Skipped wherever oracle/_ref was not built."""
import numpy as np
import pytest

from ldpc_testlib import (ASP_DEC, BP_DEC, IMS_DEC, LMS_DEC, MS_DEC, SP_DEC, TASP_DEC, Oracle, Reference, awgn_llr, load_base_matrix, ref_lib,
                          relift)

pytestmark = pytest.mark.skipif(ref_lib() is None, reason="oracle/_ref not built (needs the upstream tree)")


@pytest.mark.parametrize("dec_id,M,snr,frames,maxiter,seed", [
    (MS_DEC, 64, 1.5, 60, 50, 11), (MS_DEC, 64, 2.5, 60, 10, 12), (MS_DEC, 7, 3.0, 60, 30, 13),
    (MS_DEC, 126, 1.7, 12, 50, 14), (LMS_DEC, 64, 1.2, 40, 50, 15), (LMS_DEC, 200, 1.4, 8, 50, 16),
    (SP_DEC, 64, 1.5, 40, 50, 17), (SP_DEC, 33, 2.0, 30, 25, 18), (IMS_DEC, 64, 2.2, 30, 50, 19),
    (BP_DEC, 64, 1.3, 40, 40, 33), (BP_DEC, 126, 1.7, 8, 30, 34), (BP_DEC, 1, 4.0, 300, 20, 35), (BP_DEC, 5, 2.0, 60, 30, 36),
    (ASP_DEC, 64, 1.4, 30, 40, 30), (ASP_DEC, 126, 1.7, 8, 30, 31), (ASP_DEC, 5, 3.0, 40, 30, 32),
    (TASP_DEC, 64, 1.5, 40, 15, 20), (TASP_DEC, 126, 1.7, 10, 15, 21), (TASP_DEC, 9, 3.0, 40, 30, 22),
])
def test_restatement_equals_compiled_reference(dec_id, M, snr, frames, maxiter, seed):
    H = relift(load_base_matrix(), M)
    llr = awgn_llr(H, M, snr, seed, frames)
    llr[0, :5] = [0.0, -0.0, 25.0, -25.0, 1e-300]  # zeros, clamp range (SP INPUT_LIMIT 20), denormal-ish
    o, r = Oracle(H, M), Reference(dec_id, H, M)
    for decision in (0, 1) if dec_id != TASP_DEC else (0,):  # TASP: decision is dead upstream
        d1, i1, a1 = o.decode(dec_id, llr, maxiter, decision)
        d2, i2, a2 = r.decode(dec_id, llr, maxiter, decision)
        assert np.array_equal(i1, i2)
        assert np.array_equal(d1, d2, equal_nan=True)
        assert np.array_equal(a1, a2, equal_nan=True)  # SP clobbers its input identically


def test_irregular_small_matrix():
    # hand-made 3x6 base matrix with an empty-heavy row, a weight-1 column and shift == M-1
    H = np.array([[0, -1, 3, -1, 2, 0], [4, 1, -1, 0, -1, -1], [-1, 2, 0, 4, 4, 1]], dtype=np.int16)
    M = 5
    rng = np.random.RandomState(3)
    llr = rng.randn(50, 6 * M) * 2.0 + 1.0
    for dec_id in (MS_DEC, LMS_DEC, SP_DEC):
        o, r = Oracle(H, M), Reference(dec_id, H, M)
        d1, i1, _ = o.decode(dec_id, llr, 30, 0)
        d2, i2, _ = r.decode(dec_id, llr, 30, 0)
        assert np.array_equal(i1, i2) and np.array_equal(d1, d2)


@pytest.mark.parametrize("rh,nh,M,weights,seed", [(4, 8, 96, (3, 4, 2, 4), 1), (12, 24, 81, (3, 3, 6, 2, 11, 3), 2), (8, 20, 128, (2, 3, 8, 3, 2), 3),
                                                   (6, 15, 30, (4, 2, 5), 4)])
def test_restatement_equals_compiled_reference_on_random_protographs(rh, nh, M, weights, seed):
    """Other code shapes than the example code (the same ones the GPU suite uses): every restated decoder against the compiled
    upstream one."""
    from ldpc_testlib import random_qc_code
    H = random_qc_code(np.random.RandomState(seed), rh, nh, M, weights)
    llr = np.concatenate([awgn_llr(H, M, s, 200 + i, 10) for i, s in enumerate((1.5, 3.0, 4.5))])
    for dec_id in (MS_DEC, LMS_DEC, IMS_DEC, SP_DEC, ASP_DEC, TASP_DEC, BP_DEC):
        d1, i1, a1 = Oracle(H, M).decode(dec_id, llr, 25, 0)
        d2, i2, a2 = Reference(dec_id, H, M).decode(dec_id, llr, 25, 0)
        assert np.array_equal(i1, i2) and np.array_equal(d1, d2, equal_nan=True) and np.array_equal(a1, a2, equal_nan=True), dec_id


@pytest.mark.parametrize("rh,nh,M,seed", [(4, 8, 64, 1), (3, 7, 33, 2), (6, 9, 126, 3), (5, 12, 1, 4)])
def test_asp_branch_for_codes_whose_columns_all_have_weight_two(rh, nh, M, seed):
    """sum_prod_gf2_decod_qc_lm takes its own branch when every block column holds exactly two circulants (asp_all_cw_2,
    decoders.cpp:1027-1044, :2431-2480: no clamping, messages formed from the channel value and the other edge directly): the
    restatement against the compiled reference, hard decisions / return values / soft values / the clobbered input."""
    from ldpc_testlib import cycle_code
    H = cycle_code(np.random.RandomState(seed), rh, nh, M)
    assert ((H >= 0).sum(axis=0) == 2).all()
    llr = np.concatenate([awgn_llr(H, M, s, 300 + i, 12) for i, s in enumerate((0.0, 3.0, 6.0))])
    llr[0, :5] = [0.0, -0.0, 25.0, -25.0, 1e-300]
    for decision in (0, 1):
        d1, i1, a1 = Oracle(H, M).decode(ASP_DEC, llr, 25, decision)
        d2, i2, a2 = Reference(ASP_DEC, H, M).decode(ASP_DEC, llr, 25, decision)
        assert np.array_equal(i1, i2), (i1, i2)
        assert np.array_equal(d1, d2, equal_nan=True) and np.array_equal(a1, a2, equal_nan=True)
    assert (i1 > 0).any()                              # the decoder does work (not only codewords at the input)
