#!/usr/bin/env python3
"""Prints the 2^(k/128) table used by ldpc_spec::exp_glibc (ldpc-lib_amd/csrc/ldpc_spec.hpp) -- the table of the exp()
algorithm of glibc >= 2.28 / ARM optimized-routines (N = 128): for k = 0..127
    T[2k+1] = bits(H_k) - (k << 45),  H_k = 2^(k/128) rounded to nearest double
    T[2k]   = bits(tail_k),           tail_k = (2^(k/128) / H_k - 1) rounded to nearest double
computed here with 80-digit decimal arithmetic.  tests/test_host_cpu.py checks that a C transcription of the algorithm with this
table returns libm's exp() bit for bit on this host."""
import struct
from decimal import Decimal, getcontext

getcontext().prec = 80


def bits(x):
    return struct.unpack("<Q", struct.pack("<d", x))[0]


def table():
    t = []
    for k in range(128):
        exact = Decimal(2) ** (Decimal(k) / Decimal(128))
        h = float(exact)                                  # correctly rounded
        tail = float(exact / Decimal(h) - 1)
        t += [bits(tail), (bits(h) - (k << 45)) & 0xFFFFFFFFFFFFFFFF]
    return t


if __name__ == "__main__":
    t = table()
    for i in range(0, 256, 4):
        print("    " + ", ".join("0x%016xull" % v for v in t[i:i + 4]) + ",")
