#!/usr/bin/env python3
"""Record the interleaver maps of the COMPILED UPSTREAM code (oracle/_ref, direct_inverse_perm.cpp) as a fixture:
tests/golden/interleavers.npz.  Build container only (needs `make -C oracle ref`).  Data only: configurations and index maps."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from ldpc_testlib import GOLDEN_DIR, load_base_matrix, ref_lib, relift  # noqa: E402


def configs():
    H0 = load_base_matrix()
    mats = {"a": H0, "b": H0[:, :30], "c": H0[:13, :27], "d": np.hstack([H0, H0[:, 5:6]])}   # 32, 30, 27 and 33 block columns
    out = []
    for name, H in mats.items():
        for M in ((7, 8) if name in "ab" else (8, 9)):
            Hm = relift(H, M) if H.shape == H0.shape else np.where(H < 0, -1, H % M).astype(np.int16)
            for h in (1, 2, 3, 4):
                for mode, bs, st in ((0, 0, 0), (1, 0, 0), (2, 0, 0), (3, 50, 0), (3, 16, 0), (4, 0, 1)):
                    out.append((name, Hm, M, h, mode, bs, st))
                N = Hm.shape[1] * M
                for st in (2, 3, 4, 8):
                    if N % st == 0:
                        out.append((name, Hm, M, h, 4, 0, st))
    return out


def main():
    lib = ref_lib()
    if lib is None:
        sys.exit("oracle/_ref/libldpc_ref.so missing: run `make -C oracle ref`")
    data = {}
    n = 0
    for name, H, M, h, mode, bs, st in configs():
        b, c = H.shape
        N = c * M
        if mode == 2 and c % h:      # upstream's deterministic mode needs room for halfmlog column groups per maximal-weight column
            cw = (H >= 0).sum(axis=0)
            mrp = int((cw == cw.max()).sum())
            if c - h * mrp < 0:
                continue
        Hs = np.ascontiguousarray(H, dtype=np.int16)
        d = np.zeros(N, dtype=np.int32)
        i = np.zeros(N, dtype=np.int32)
        rc = lib.ref_perm_maps(b, c, M, 1 << (2 * h), h, mode, bs, st, Hs.ctypes.data_as(C.c_void_p), d.ctypes.data_as(C.c_void_p),
                               i.ctypes.data_as(C.c_void_p))
        assert rc == 0
        if sorted(d.tolist()) != list(range(N)):
            print(f"skip (upstream's map is not a permutation): {name} M={M} h={h} mode={mode} bs={bs} st={st}")
            continue
        key = f"{n:03d}"
        data[key + "_cfg"] = np.array([M, h, mode, bs, st], dtype=np.int32)
        data[key + "_H"] = Hs
        data[key + "_direct"] = d.astype(np.int16)
        data[key + "_inverse"] = i.astype(np.int16)
        n += 1
    np.savez_compressed(os.path.join(GOLDEN_DIR, "interleavers.npz"), **data)
    print(n, "configurations")


if __name__ == "__main__":
    main()
