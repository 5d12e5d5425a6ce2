"""Shared test helpers: fixtures, ctypes bindings to the CPU oracle (oracle/liboracle.so) and, when it was
built in the container that has the upstream tree, to the compiled reference (oracle/_ref/libldpc_ref.so).

Everything here is test infrastructure; the product (ldpc-lib_amd) never imports it.
"""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")

# decoders.h:16-28 enum DEC_ID
BP_DEC, SP_DEC, ASP_DEC, MS_DEC, IMS_DEC, IASP_DEC, FHT_DEC, TASP_DEC, LMS_DEC, LCHE_DEC = range(10)

c_double_p = C.POINTER(C.c_double)
c_int_p = C.POINTER(C.c_int)
c_short_p = C.POINTER(C.c_short)


def load_base_matrix():
    """the 16x32 base matrix the reference's own `search` produced (shifts mod 126)."""
    return np.loadtxt(os.path.join(GOLDEN_DIR, "h16x32_m126.txt"), dtype=np.int32)


def relift(H, M):
    """main_simulation.cpp:400-414: entries > 0 become entry % M; in column rows-1 a result of 0 becomes 1."""
    H = np.array(H, dtype=np.int32, copy=True)
    rows = H.shape[0]
    for i in range(H.shape[0]):
        for j in range(H.shape[1]):
            if H[i, j] > 0:
                t = int(H[i, j]) % M
                if j == rows - 1 and t == 0:
                    t = 1
                H[i, j] = t
    return H


def _as_double_p(a):
    return a.ctypes.data_as(c_double_p)


class SimResult(C.Structure):
    _fields_ = [("ber", C.c_double), ("fer", C.c_double), ("nse", C.c_longlong), ("nde", C.c_longlong),
                ("nue", C.c_longlong), ("experiment", C.c_longlong), ("sum_abs_iter", C.c_longlong),
                ("rng_next", C.c_uint)]


_oracle_lib = None


def oracle_lib():
    """Load (building on demand with plain gcc/g++) the CPU restatement."""
    global _oracle_lib
    if _oracle_lib is not None:
        return _oracle_lib
    so = os.path.join(ORACLE_DIR, "liboracle.so")
    srcs = [os.path.join(ORACLE_DIR, f) for f in ("ldpc_oracle.c", "ldpc_oracle.h", "harness_oracle.cpp", "harness_oracle.h")]
    if not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        subprocess.check_call(["make", "-s", "-C", ORACLE_DIR, "liboracle.so"])
    lib = C.CDLL(so)
    lib.orc_open.restype = C.c_void_p
    lib.orc_open.argtypes = [C.c_int, C.c_int, C.c_int, c_short_p]
    lib.orc_close.argtypes = [C.c_void_p]
    lib.orc_n.argtypes = [C.c_void_p]
    lib.orc_min_sum.argtypes = [C.c_void_p, c_double_p, c_double_p, C.c_int, C.c_int, C.c_double]
    lib.orc_lmin_sum.argtypes = [C.c_void_p, c_double_p, c_double_p, C.c_int, C.c_int]
    lib.orc_sum_prod.argtypes = [C.c_void_p, c_double_p, c_double_p, C.c_int, C.c_int]
    lib.orc_imin_sum.argtypes = [C.c_void_p, c_double_p, c_double_p, C.c_int, C.c_int, C.c_double, C.c_double, C.c_int, C.c_int]
    lib.orc_sum_prod_gf2.argtypes = [C.c_void_p, c_double_p, c_double_p, C.c_int, C.c_int]
    lib.orc_bp.argtypes = [C.c_void_p, c_double_p, c_double_p, C.c_int, C.c_int]
    lib.orc_bp_stale.argtypes = [C.c_void_p]
    lib.orc_bp_stale.restype = C.POINTER(C.c_ubyte)
    lib.orc_tdmp_sum_prod.argtypes = [C.c_void_p, c_double_p, c_double_p, C.c_int, c_double_p]
    lib.orc_syndrome_nonzero.argtypes = [C.c_void_p, c_double_p]
    lib.orc_qam_modulate.argtypes = [C.c_int, c_double_p, C.c_int, c_double_p]
    lib.orc_qam_demodulate.argtypes = [C.c_int, C.c_double, C.c_double, c_double_p, C.c_int, c_double_p, C.c_int]
    lib.orc_bp_simulation.argtypes = [C.c_int, C.c_int, c_int_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double,
                                      C.c_double, C.c_int, C.c_int, C.c_int, C.c_uint, C.POINTER(SimResult), c_int_p]
    lib.orc_rng_gaussians.argtypes = [C.c_uint, C.c_int, c_double_p, C.c_int]
    lib.orc_awgn_llr.argtypes = [C.c_uint, C.c_int, C.c_double, C.c_double, c_double_p, C.c_int]
    _oracle_lib = lib
    return lib


def ref_lib():
    """The compiled upstream reference (only exists where oracle/Makefile's `ref` target was run). None if absent."""
    so = os.path.join(ORACLE_DIR, "_ref", "libldpc_ref.so")
    if not os.path.exists(so):
        return None
    lib = C.CDLL(so)
    lib.ref_open.restype = C.c_void_p
    lib.ref_open.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, c_short_p]
    lib.ref_close.argtypes = [C.c_void_p]
    lib.ref_decode.argtypes = [C.c_void_p, C.c_int, c_double_p, c_double_p, C.c_int, C.c_int]
    lib.ref_qam_modulate.argtypes = [C.c_int, c_double_p, C.c_int, c_double_p]
    lib.ref_qam_demodulate.argtypes = [C.c_int, C.c_double, C.c_double, c_double_p, C.c_int, c_double_p, C.c_int]
    return lib


class _Decoder:
    """Common batch front end: decode(llr[B,N]) -> (decword[B,N] float64, iters[B] int32, llr_after[B,N])."""

    def decode(self, dec_id, llr, maxiter, decision=0):
        llr = np.ascontiguousarray(llr, dtype=np.float64)
        single = llr.ndim == 1
        if single:
            llr = llr[None, :]
        B, N = llr.shape
        assert N == self.N
        dec = np.empty((B, N), dtype=np.float64)
        its = np.empty(B, dtype=np.int32)
        after = llr.copy()
        for b in range(B):
            its[b] = self._one(dec_id, after[b], dec[b], maxiter, decision)
        if single:
            return dec[0], int(its[0]), after[0]
        return dec, its, after


class Oracle(_Decoder):
    def __init__(self, H, M):
        self.lib = oracle_lib()
        H = np.ascontiguousarray(H, dtype=np.int16)
        self.rh, self.nh = H.shape
        self.M = M
        self.N = self.nh * M
        self.R = self.rh * M
        self.h = self.lib.orc_open(self.rh, self.nh, M, H.ctypes.data_as(c_short_p))
        assert self.h

    def close(self):
        if self.h:
            self.lib.orc_close(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _one(self, dec_id, y, dec, maxiter, decision):
        yp, dp = _as_double_p(y), _as_double_p(dec)
        if dec_id == MS_DEC:
            return self.lib.orc_min_sum(self.h, yp, dp, maxiter, decision, 0.8)
        if dec_id == LMS_DEC:
            return self.lib.orc_lmin_sum(self.h, yp, dp, maxiter, decision)
        if dec_id == SP_DEC:
            return self.lib.orc_sum_prod(self.h, yp, dp, maxiter, decision)
        if dec_id == IMS_DEC:
            return self.lib.orc_imin_sum(self.h, yp, dp, maxiter, decision, 0.8, 1.4, 6, 8)
        if dec_id == BP_DEC:
            return self.lib.orc_bp(self.h, yp, dp, maxiter, decision)
        if dec_id == ASP_DEC:
            return self.lib.orc_sum_prod_gf2(self.h, yp, dp, maxiter, decision)
        if dec_id == TASP_DEC:  # decision is dead upstream: decword is always hard; decision=1 here returns the posteriors
            if decision:
                tmp = np.empty_like(dec)
                return self.lib.orc_tdmp_sum_prod(self.h, yp, _as_double_p(tmp), maxiter, dp)
            return self.lib.orc_tdmp_sum_prod(self.h, yp, dp, maxiter, None)
        raise ValueError(dec_id)


class Reference(_Decoder):
    """The compiled upstream decoders (oracle/_ref). Construct only when ref_lib() is not None."""

    def __init__(self, dec_id, H, M):
        self.lib = ref_lib()
        assert self.lib is not None
        H = np.ascontiguousarray(H, dtype=np.int16)
        self.rh, self.nh = H.shape
        self.M = M
        self.N = self.nh * M
        self.dec_id = dec_id
        self.h = self.lib.ref_open(dec_id, self.rh, self.nh, M, H.ctypes.data_as(c_short_p))
        assert self.h

    def close(self):
        if self.h:
            self.lib.ref_close(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _one(self, dec_id, y, dec, maxiter, decision):
        assert dec_id == self.dec_id
        return self.lib.ref_decode(self.h, dec_id, _as_double_p(y), _as_double_p(dec), maxiter, decision)


def code_rate(H):
    rh, nh = H.shape
    return (nh - rh) / nh


def awgn_llr(H, M, snr_db, seed, frames, burn_codeword=True):
    """LLRs in the reference's draw order (bp_simulation.cpp:512,600-605): one mt19937 stream, the
    (nh-rh)*M next_random_int draws of random_codeword() first, then frames*N Gaussians."""
    lib = oracle_lib()
    rh, nh = H.shape
    N = nh * M
    out = np.empty(frames * N, dtype=np.float64)
    lib.orc_awgn_llr(seed, (nh - rh) * M if burn_codeword else 0, snr_db, code_rate(H), _as_double_p(out), frames * N)
    return out.reshape(frames, N)


def pack_bits(dec):
    """decword (0.0/1.0 doubles, [B,N]) -> little-endian packed uint32 words [B, ceil(N/32)] (bit i of word w = variable 32*w+i)."""
    dec = np.asarray(dec)
    B, N = dec.shape
    W = (N + 31) // 32
    bits = np.zeros((B, W * 32), dtype=np.uint8)
    bits[:, :N] = dec != 0
    return np.packbits(bits.reshape(B, W, 32), axis=2, bitorder="little").view(np.uint32).reshape(B, W)


# ---- CPU twin of the device noise generator (ldpc-lib_amd/csrc/ldpc_frontend.hpp) ---------------------------------
def philox4x32_10(c0, c1, c2, c3, k0, k1):
    """Vectorised Philox4x32-10 (Salmon et al. 2011).  All arguments uint32 arrays (broadcastable)."""
    c = [np.asarray(x, dtype=np.uint64) for x in np.broadcast_arrays(c0, c1, c2, c3)]
    k0 = np.uint64(k0)
    k1 = np.uint64(k1)
    M0, M1, mask = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57), np.uint64(0xFFFFFFFF)
    for _ in range(10):
        p0 = M0 * c[0]
        p1 = M1 * c[2]
        n0 = ((p1 >> np.uint64(32)) ^ c[1] ^ k0) & mask
        n1 = p1 & mask
        n2 = ((p0 >> np.uint64(32)) ^ c[3] ^ k1) & mask
        n3 = p0 & mask
        c = [n0, n1, n2, n3]
        k0 = (k0 + np.uint64(0x9E3779B9)) & mask
        k1 = (k1 + np.uint64(0xBB67AE85)) & mask
    return [x.astype(np.uint32) for x in c]


def philox_gauss_pairs(seed, frames, npairs, tag=0):
    """Box-Muller pairs exactly as gauss_pair() on the device (up to libm-vs-ocml rounding): returns [len(frames), 2*npairs]."""
    frames = np.asarray(frames, dtype=np.uint64)[:, None]
    pair = np.arange(npairs, dtype=np.uint64)[None, :]
    x = philox4x32_10(frames & np.uint64(0xFFFFFFFF), frames >> np.uint64(32), pair, np.uint64(tag),
                      np.uint64(seed & 0xFFFFFFFF), np.uint64(seed >> 32))
    a = ((x[0].astype(np.uint64) << np.uint64(32)) | x[1].astype(np.uint64)) >> np.uint64(11)
    b = ((x[2].astype(np.uint64) << np.uint64(32)) | x[3].astype(np.uint64)) >> np.uint64(11)
    u1 = (a.astype(np.float64) + 0.5) / 9007199254740992.0
    u2 = (b.astype(np.float64) + 0.5) / 9007199254740992.0
    rad = np.sqrt(-2.0 * np.log(u1))
    g = np.empty((frames.shape[0], 2 * npairs))
    g[:, 0::2] = rad * np.cos(2.0 * np.pi * u2)
    g[:, 1::2] = rad * np.sin(2.0 * np.pi * u2)
    return g


def bpsk_sigma(H, snr_db, punctured_blocks=0):
    rh, nh = H.shape
    rate = (nh - rh) / (nh - punctured_blocks)
    return np.sqrt(10.0 ** (-snr_db / 10.0) / 2 / rate)   # bp_simulation.cpp:444-445


def syndrome_np(H, M, hard_bits):
    """hard_bits [B,N] 0/1 -> [B] bool: any parity check fails (edge rule: check (j,n) -- variable (k,(n+c)%M))."""
    hard_bits = np.asarray(hard_bits, dtype=np.uint8)
    B = hard_bits.shape[0]
    rh, nh = H.shape
    cols = hard_bits.reshape(B, nh, M)
    fail = np.zeros(B, dtype=bool)
    for j in range(rh):
        s = np.zeros((B, M), dtype=np.uint8)
        for k in range(nh):
            if H[j, k] >= 0:
                s ^= np.roll(cols[:, k, :], -int(H[j, k]) % M, axis=1)
        fail |= s.any(axis=1)
    return fail


def unpack_bits(hard_words, N):
    """packed uint32 [B,W] -> [B,N] uint8"""
    w = np.ascontiguousarray(hard_words).view(np.uint32)
    bits = np.unpackbits(w.view(np.uint8).reshape(w.shape[0], -1), axis=1, bitorder="little")
    return bits[:, :N]


def random_qc_code(rng, rh, nh, M, info_weight):
    """A random protograph of the usual shape: dual-diagonal parity part (block columns 0..rh-1, shifts 0, one extra entry
    closing the chain) and `info_weight[k]` random circulants in every information column."""
    H = -np.ones((rh, nh), dtype=np.int16)
    for j in range(rh):
        H[j, j] = 0
        if j + 1 < rh:
            H[j + 1, j] = 0
    H[0, rh - 1] = 1 % M
    H[rh // 2, rh - 1] = 0 if H[rh // 2, rh - 1] < 0 else H[rh // 2, rh - 1]
    for k in range(rh, nh):
        rows = rng.choice(rh, size=min(rh, info_weight[(k - rh) % len(info_weight)]), replace=False)
        for j in rows:
            H[j, k] = rng.randint(0, M)
    for j in range(rh):                      # every check sees at least two information columns
        while (H[j, rh:] >= 0).sum() < 2:
            H[j, rh + rng.randint(0, nh - rh)] = rng.randint(0, M)
    return H


def cycle_code(rng, rh, nh, M):
    """A protograph in which EVERY block column has exactly two circulants (and every block row at least two): the shape upstream's
    sum_prod_gf2_decod_qc_lm decodes in its own branch (asp_all_cw_2, decoders.cpp:1027-1044, :2431-2480)."""
    while True:
        H = -np.ones((rh, nh), dtype=np.int16)
        for k in range(nh):
            for j in rng.choice(rh, size=2, replace=False):
                H[j, k] = rng.randint(0, M)
        if ((H >= 0).sum(axis=1) >= 2).all():
            return H


def multi_block_code(rng, M, blocks=(4, 4), ninfo=6):
    """A base matrix whose parity part consists of several dual-diagonal blocks (bp_simulation.cpp:142-191 encodes them from the last
    to the first): block q has its own double diagonal and special column (shifts 0, s > 0, 0); earlier blocks' rows also touch the
    parity columns of later blocks, which they treat like information columns."""
    b = int(sum(blocks))
    H = -np.ones((b, b + ninfo), dtype=np.int16)
    off = 0
    for rb in blocks:
        hi = off + rb
        for t in range(rb - 1):
            H[off + t, off + t] = 0
            H[off + t + 1, off + t] = 0
        H[off, hi - 1] = 0
        H[off + rb // 2, hi - 1] = 1 + rng.randint(0, max(M - 1, 1)) if M > 1 else 0
        H[hi - 1, hi - 1] = 0
        for _ in range(2):                       # rows of this block on parity columns of LATER blocks
            if hi < b:
                H[off + rng.randint(0, rb), rng.randint(hi, b)] = rng.randint(0, M)
        off = hi
    for k in range(b, b + ninfo):
        for j in rng.choice(b, size=3, replace=False):
            H[j, k] = rng.randint(0, M)
    for j in range(b):
        while (H[j, b:] >= 0).sum() < 1:
            H[j, b + rng.randint(0, ninfo)] = rng.randint(0, M)
    return H
