"""GPU: code shapes.  (a) shapes of upstream's own scenario files that the resident kernels must take without scratch surprises
(30 x 60 at lifting 67, files/input12L.jsonx:3-6; row weight 16); (b) shapes beyond the LDS-resident kernels' limits, which run on
the shape-unlimited tier (ldpc_global.hpp) instead of being refused -- upstream's decod_open has no such limits
(decoders.cpp:348-791); (c) that tier forced onto the golden vectors of the compiled reference.  Everything bit for bit."""
import os

import numpy as np
import pytest

from ldpc_testlib import (ASP_DEC, BP_DEC, GOLDEN_DIR, IMS_DEC, LMS_DEC, MS_DEC, SP_DEC, TASP_DEC, Oracle, awgn_llr, load_base_matrix, pack_bits, random_qc_code,
                          relift)

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


def _check(L, torch, dec_id, H, M, llr, maxiter, expect_kernel=None, soft=True):
    o = Oracle(H, M)
    with L.LdpcHip(dec_id, H, M) as dec:
        if expect_kernel:
            assert expect_kernel in dec.kernel_name, dec.kernel_name
        hard, iters, sv = dec.decode(torch.from_numpy(llr).cuda(), maxiter, want_soft=soft)
        torch.cuda.synchronize()
        d_ref, it_ref, _ = o.decode(dec_id, llr, maxiter, 0)
        assert np.array_equal(iters.cpu().numpy(), it_ref)
        assert np.array_equal(hard.cpu().numpy().view(np.uint32), pack_bits(d_ref))
        if soft and dec_id != TASP_DEC:
            s_ref, _, _ = Oracle(H, M).decode(dec_id, llr, maxiter, 1)       # a fresh state: Gallager BP carries its syndrome from frame to frame
            assert np.array_equal(sv.cpu().numpy(), s_ref, equal_nan=True)
        return dec.kernel_name, it_ref


def _rows_of_weight(rng, rh, nh, M, target):
    """Dual-diagonal parity part + information part in which every block row has exactly `target` circulants in total and every
    information column at least one."""
    H = random_qc_code(rng, rh, nh, M, [1])
    H[:, rh:] = -1
    ninfo = nh - rh
    start = 0
    for j in range(rh):
        need = target - int((H[j] >= 0).sum())
        assert 2 <= need <= ninfo
        for q in range(need):                              # a rolling window: rows overlap, all columns get covered
            H[j, rh + (start + q) % ninfo] = rng.randint(0, M)
        start += need - 1
    assert ((H >= 0).sum(axis=1) == target).all() and ((H[:, rh:] >= 0).sum(axis=0) >= 1).all()
    return H


def _llr(H, M, snr, seed, frames):
    return awgn_llr(np.asarray(H, dtype=np.int32), M, snr, seed, frames, burn_codeword=False)


@pytest.mark.parametrize("dec_id,maxiter,snr", [(MS_DEC, 50, 2.2), (LMS_DEC, 50, 1.4), (TASP_DEC, 15, 1.4)])
def test_thirty_by_sixty_at_lifting_67(L, torch, dec_id, maxiter, snr):
    """The shape of files/input12L.jsonx (rows 30, columns 60, tailbite length 67; information column weights 2 / 3 / 16).
    All three run on hiprtc instances of the resident bodies.  TDMP sum-product has 206 circulants: more than the 144 whose state one
    lane per check could hold in registers (rounds 1-2: shape-unlimited tier), but with two lanes per check a lane holds half a row
    and the resident body takes it at one wave per SIMD."""
    rng = np.random.RandomState(67)
    H = random_qc_code(rng, 30, 60, 67, [2, 3, 3, 16, 2, 3])
    llr = _llr(H, 67, snr, 5, 24)
    name, it = _check(L, torch, dec_id, H, 67, llr, maxiter, expect_kernel="hiprtc")
    assert (it > 0).any() and (it < 0).any(), it       # both converged and failed frames in the sample


@pytest.mark.parametrize("dec_id", [MS_DEC, LMS_DEC])
def test_row_weight_sixteen(L, torch, dec_id):
    H = _rows_of_weight(np.random.RandomState(16), 4, 24, 64, 16)
    _check(L, torch, dec_id, H, 64, _llr(H, 64, 3.0, 7, 32), 30)


@pytest.mark.parametrize("rh,nh,M,target", [(3, 9, 33, 5), (4, 12, 47, 5), (5, 15, 95, 6), (4, 16, 20, 6), (6, 18, 255, 7), (8, 24, 129, 9), (3, 8, 64, 0),
                                            (5, 11, 32, 0), (7, 15, 31, 0), (16, 32, 256, 0), (10, 36, 250, 0)])
def test_tdmp_with_two_lanes_per_check_on_odd_shapes(L, torch, rh, nh, M, target):
    """tasp_body splits every check row over two lanes (first half ascending, second half descending, an odd row padded with a factor
    1.0) and pairs lane l with lane l + 32: rows of even and odd weight (halves of 3 / 2, 3 / 3, 4 / 3, 5 / 4), liftings that
    fill a wave's halves unevenly (33, 47, 95, 129, 255), one that leaves a whole half-wave idle (20, 31) and one that fills it
    exactly (32, 64); target 0 = random column weights 2 / 3 (rows of mixed weight).  The last two have LDS images beyond 64 KB
    (N = 8192 and 9000): the packed 16-bit addresses are then 8-byte word indices."""
    rng = np.random.RandomState(rh * 100 + M)
    H = _rows_of_weight(rng, rh, nh, M, target) if target else random_qc_code(rng, rh, nh, M, [2, 3])
    name, it = _check(L, torch, TASP_DEC, H, M, _llr(H, M, 2.5, 9, 40), 15, expect_kernel="tasp_body")
    assert (it != 0).any()


@pytest.mark.parametrize("dec_id,w,maxiter", [(SP_DEC, 10, 30), (SP_DEC, 12, 30), (SP_DEC, 13, 30), (SP_DEC, 16, 30), (TASP_DEC, 12, 15), (TASP_DEC, 16, 15),
                                               (ASP_DEC, 12, 30), (ASP_DEC, 16, 30)])
def test_heavy_rows_in_the_sum_product_family(L, torch, dec_id, w, maxiter):
    """Row weights the soak does not reach (it stops at 8).  sp_body divides without the scaling / fix-up instructions up to row
    weight 12 (the range argument is about products of that many messages) and with the compiler's division beyond; TDMP halves of
    6 and 8 edges per lane; at 2 dB and at 6 dB (saturated messages, the poles of (1+A)/(1-A))."""
    H = _rows_of_weight(np.random.RandomState(100 + w), 4, 24, 64, w)
    llr = np.concatenate([_llr(H, 64, 2.0, 7, 24), _llr(H, 64, 6.0, 8, 24)])
    name, it = _check(L, torch, dec_id, H, 64, llr, maxiter, expect_kernel="hiprtc")
    assert (it > 0).any()


GLOBAL_SHAPES = [
    # what the resident kernels refuse                                   rh  nh   M    weights        snr
    ("lifting 600 > 512",                                                 16, 32, 600, None,          1.6),
    ("N * 8 B = 256 KiB > 160 KiB of LDS (lifting 1024)",                 16, 32, 1024, None,         1.6),
    ("70 block rows > 64",                                                70, 140, 8,  [2, 3, 3, 4],  2.5),
    ("row weight 20 > 16",                                                4, 28, 32,   [3],           4.0),
    ("an empty block column",                                             6, 12, 64,   [2, 3],        3.0),
]


@pytest.mark.parametrize("dec_id", [MS_DEC, LMS_DEC, TASP_DEC, SP_DEC, IMS_DEC, ASP_DEC, BP_DEC])
@pytest.mark.parametrize("why,rh,nh,M,weights,snr", GLOBAL_SHAPES)
def test_shapes_beyond_the_resident_kernels_run_on_the_global_tier(L, torch, dec_id, why, rh, nh, M, weights, snr):
    rng = np.random.RandomState(rh * 1000 + nh)
    if weights is None:
        H = relift(load_base_matrix(), M)
    elif "row weight" in why:
        H = _rows_of_weight(rng, rh, nh, M, 20)
    else:
        H = random_qc_code(rng, rh, nh, M, weights)
    if "empty" in why:
        H[:, nh - 1] = -1                               # an information column no check looks at: soft stays y
        for j in range(rh):
            while (H[j, rh:] >= 0).sum() < 2:
                H[j, rh + rng.randint(0, nh - rh - 1)] = rng.randint(0, M)
    frames = 6 if M >= 600 else 16
    # (an empty column only rules out the code-specialised bodies: the table-driven kernels of min-sum / layered min-sum take it)
    expect = None if ("empty" in why and dec_id in (MS_DEC, LMS_DEC, IMS_DEC)) else "_global_kernel"
    if dec_id == SP_DEC and M < 600:
        expect = None        # the table-driven sum-product kernel has no block-row / row-weight limits: it takes the small shapes itself
    if dec_id in (ASP_DEC, BP_DEC) and M < 600 and "empty" not in why and "70 block" not in why:
        expect = None        # small enough for the LDS-resident bodies (hiprtc instance)
    name, it = _check(L, torch, dec_id, H, M, _llr(H, M, snr, 3, frames), 15 if dec_id == TASP_DEC else 30, expect_kernel=expect)


def test_what_the_global_tier_cannot_take_still_fails_loudly(L, torch):
    """Decoders 2 and 7 need two circulants per block row (map_bin reads past a one-element row upstream): error, no fallback."""
    H1 = -np.ones((3, 6), dtype=np.int16)
    H1[0, 0] = 0; H1[1, 1] = 0; H1[2, 2] = 0; H1[1, 3] = 5; H1[2, 4] = 7; H1[1, 5] = 2; H1[2, 5] = 3  # block row 0 has a single circulant
    with pytest.raises(L.LdpcHipError):
        L.LdpcHip(TASP_DEC, H1, 600)


@pytest.mark.parametrize("name", ["ms_m64_1p2", "ms_m126_1p7", "ms_m1_4p0", "ms_m512_1p6", "lms_m64_0p8", "lms_m512_1p0", "lms_m1_4p0",
                                  "tasp_m64_1p7", "tasp_m126_1p7", "tasp_m1_4p0", "sp_m64_1p2", "sp_m64_2p0", "sp_m1_4p0", "ims_m64_2p0",
                                  "asp_m64_1p2", "asp_m128_1p7", "asp_cw2_m64_2p0", "bp_m64_2p0", "bp_m128_1p7", "bp_m64_1p0_stale"])
def test_global_tier_on_the_compiled_references_vectors(L, torch, name, monkeypatch):
    """LDPC_HIP_FORCE_GLOBAL=1: the tier takes shapes the resident kernels normally serve, so it can be pinned by the golden
    vectors the compiled upstream code produced: hard bits, return values, soft values."""
    monkeypatch.setenv("LDPC_HIP_FORCE_GLOBAL", "1")
    g = np.load(os.path.join(GOLDEN_DIR, name + ".npz"))
    H, M, dec_id, maxiter = g["H"], int(g["M"]), int(g["dec_id"]), int(g["maxiter"])
    with L.LdpcHip(dec_id, H, M) as dec:
        assert "_global_kernel" in dec.kernel_name
        hard, iters, _ = dec.decode(torch.from_numpy(g["llr"]).cuda(), maxiter)
        ns = g["soft"].shape[0]
        _, it2, soft = dec.decode(torch.from_numpy(g["llr"][:ns]).cuda(), maxiter, want_soft=True)
        torch.cuda.synchronize()
        assert np.array_equal(iters.cpu().numpy(), g["iters"]) and np.array_equal(it2.cpu().numpy(), g["iters"][:ns])
        assert np.array_equal(hard.cpu().numpy().view(np.uint32), g["hard"])
        if dec_id == TASP_DEC:   # upstream ignores `decision` here (decoders.cpp:2737-2738): the golden "soft" is the hard decision again
            assert np.array_equal((soft.cpu().numpy() > 0.5).astype(np.float64), g["soft"])
        else:
            assert np.array_equal(soft.cpu().numpy(), g["soft"])


# ---- a bounded slice of tools/soak.py inside the suite (the random-protograph evidence was builder-run only) ------------
def _soak_case(rng, big):
    rh = int(rng.randint(2, 30 if big else 13))
    nh = rh + int(rng.randint(2, 30 if big else 14))
    M = int(rng.choice([1, 2, 3, 5, 8, 13, 16, 21, 27, 32, 33, 40, 47, 48, 63, 64, 65, 67, 96, 100, 126, 128, 129, 160, 200, 256] + ([300, 513] if big else [])))
    weights = tuple(int(x) for x in rng.randint(2, min(rh, 6) + 1, size=4))
    return rh, nh, M, random_qc_code(rng, rh, nh, M, weights)


def test_soak_slice_random_protographs_on_the_shape_unlimited_tier(L, torch, monkeypatch):
    """Fixed seeds, <= 60 s: random protographs (2..29 block rows, liftings 1..513) x all seven decoders on the shape-unlimited tier
    (no hiprtc, so the time goes into decoding) against the oracle -- hard bits, return values and soft values bit for bit."""
    import time
    monkeypatch.setenv("LDPC_HIP_FORCE_GLOBAL", "1")
    rng = np.random.RandomState(20261005)
    t0, done = time.time(), 0
    for case in range(40):
        if time.time() - t0 > 45 and done >= 42:
            break
        rh, nh, M, H = _soak_case(rng, True)
        frames = 6 if M * nh > 20000 else 16 if M * nh > 2000 else 40
        llr = np.concatenate([awgn_llr(H, M, s, 900 + case, frames // 2) for s in (2.0, 5.0)])
        llr[0, :3] = [0.0, -0.0, 40000.0]
        for dec_id in (MS_DEC, LMS_DEC, IMS_DEC, SP_DEC, TASP_DEC, ASP_DEC, BP_DEC):
            name, _ = _check(L, torch, dec_id, H, M, llr, 30, expect_kernel="global")
            done += 1
    assert done >= 42, done


def test_soak_slice_random_protographs_on_hiprtc_instances(L, torch, tmp_path, monkeypatch):
    """The same against code-specialised instances compiled on the spot (hiprtc): five random codes x the three min-sum bodies, with
    lifting classes that hit ms_small / ms_m64 / ms_chunk / ms_body."""
    monkeypatch.setenv("LDPC_HIP_CACHE_DIR", str(tmp_path))
    rng = np.random.RandomState(20261006)
    done = 0
    for case, M in enumerate((13, 64, 100, 27, 200)):
        rh = int(rng.randint(3, 9))
        nh = rh + int(rng.randint(3, 10))
        H = random_qc_code(rng, rh, nh, M, tuple(int(x) for x in rng.randint(2, min(rh, 5) + 1, size=4)))
        if (H >= 0).sum(axis=1).max() > 8:
            continue
        llr = np.concatenate([awgn_llr(H, M, s, 950 + case, 12) for s in (2.0, 5.0)])
        llr[0, :3] = [0.0, -0.0, 40000.0]
        for dec_id in (MS_DEC, LMS_DEC, IMS_DEC):
            name, _ = _check(L, torch, dec_id, H, M, llr, 30)
            assert "hiprtc" in name, name
            done += 1
    assert done >= 9


def test_chain_changes_do_not_touch_the_shape_unlimited_tiers_workspace(L, torch, monkeypatch):
    """Regression (round 2, 4209a7b): prepare_chain -- run whenever the modulation, the interleaver or the codewords of a context
    change -- freed the shape-unlimited tier's message workspace and the next decode used the stale pointer.  Decode, change the
    chain twice with allocations in between that would recycle a freed block, decode again: still the oracle's bits."""
    monkeypatch.setenv("LDPC_HIP_FORCE_GLOBAL", "1")
    M = 64
    H = relift(load_base_matrix(), M)
    llr = awgn_llr(H, M, 1.5, 77, 48)
    x = torch.from_numpy(llr).cuda()
    for dec_id in (BP_DEC, MS_DEC):
        o = Oracle(H, M)
        d_ref, it_ref, _ = o.decode(dec_id, llr, 20, 0)
        with L.LdpcHip(dec_id, H, M) as dec:
            assert "global" in dec.kernel_name
            if dec_id == BP_DEC:
                dec.set_bp_chain(True, True)
            hard, iters, _ = dec.decode(x, 20)
            assert np.array_equal(iters.cpu().numpy(), it_ref)
            junk = []
            for mod in (1, 0, 2):
                dec.awgn_llr(2.0, 1, 0, 8, modulation=mod)                  # prepare_chain(mod)
                junk.append(torch.full((1 << 20,), float("nan"), dtype=torch.float64, device="cuda"))
            dec.set_codewords(np.zeros((2, dec.N), dtype=np.uint8))
            dec.awgn_llr(2.0, 1, 0, 8)
            junk.append(torch.full((1 << 22,), float("nan"), dtype=torch.float64, device="cuda"))
            torch.cuda.synchronize()
            if dec_id == BP_DEC:
                dec.set_bp_chain(True, True)                                # same starting syndrome as the oracle's fresh state
            hard, iters, _ = dec.decode(x, 20)
            torch.cuda.synchronize()
            assert np.array_equal(iters.cpu().numpy(), it_ref)
            assert np.array_equal(hard.cpu().numpy().view(np.uint32), pack_bits(d_ref))
            del junk


@pytest.mark.parametrize("rh,nh,M,seed", [(4, 8, 64, 1), (3, 7, 33, 2), (6, 9, 126, 3), (5, 12, 1, 4), (7, 11, 300, 5)])
def test_asp_on_codes_whose_columns_all_have_weight_two(L, torch, rh, nh, M, seed):
    """sum_prod_gf2_decod_qc_lm's own branch for codes in which every block column holds exactly two circulants (asp_all_cw_2,
    decoders.cpp:1027-1044, :2431-2480): round 2 refused them; the shape-unlimited tier now runs that branch -- hard bits, return
    values and soft values equal the oracle's (which tests/test_oracle_vs_ref.py pins to the compiled reference on the same
    shapes, and tests/golden/asp_cw2_m64_2p0.npz to its recorded output)."""
    from ldpc_testlib import cycle_code
    H = cycle_code(np.random.RandomState(seed), rh, nh, M)
    llr = np.concatenate([awgn_llr(H, M, s, 300 + i, 12) for i, s in enumerate((0.0, 3.0, 6.0))])
    llr[0, :5] = [0.0, -0.0, 25.0, -25.0, 1e-300]
    name, it = _check(L, torch, ASP_DEC, H, M, llr, 25, expect_kernel="asp_global_kernel")
    assert (it > 0).any()
