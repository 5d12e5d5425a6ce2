#!/usr/bin/env python3
"""Prints the data of ldpc_spec::log_glibc (ldpc-lib_amd/csrc/ldpc_spec.hpp): the constants of the log() algorithm of
glibc >= 2.28 (sysdeps/ieee754/dbl-64/e_log.c + e_log_data.c = ARM optimized-routines log, N = 128):
    [0] ln2hi  [1] ln2lo  [2..6] poly A[0..4] (order 6, |r| < 1/256)  [7..17] poly1 B[0..10] (order 12, near 1)
    [18 + 2i] invc_i  [19 + 2i] logc_i,  i = 0..127
Unlike exp's 2^(k/128) table the (invc, logc) pairs are not given by a formula -- the authors picked each c by a search over
candidates near the centre of its subinterval (e_log_data.c) -- so this tool READS them from the host's own libm: it finds
`__log_data` in libm.so.6 by its leading (ln2hi, ln2lo) pair and checks the structure (B[0] == -0.5, invc_i * c_i ~ 1,
logc_i ~ log(c_i)).  tests/test_host_cpu.py checks that a C transcription of log_glibc with this data returns libm's log() bit for
bit on this host, and that the header holds exactly what this tool prints."""
import math
import struct
import sys

LN2HI, LN2LO = 0x3fe62e42fefa3800, 0x3d2ef35793c76730
N = 2 + 5 + 11 + 256


def find_libm():
    for p in ("/lib/x86_64-linux-gnu/libm.so.6", "/usr/lib/x86_64-linux-gnu/libm.so.6", "/lib64/libm.so.6", "/usr/lib64/libm.so.6"):
        try:
            return open(p, "rb").read()
        except OSError:
            pass
    raise SystemExit("libm.so.6 not found")


def table():
    b = find_libm()
    pat = struct.pack("<QQ", LN2HI, LN2LO)
    pos = 0
    while True:
        pos = b.find(pat, pos)
        if pos < 0:
            raise SystemExit("__log_data not found in libm.so.6 (glibc older than 2.28?)")
        vals = struct.unpack("<%dQ" % N, b[pos:pos + 8 * N])
        d = struct.unpack("<%dd" % N, b[pos:pos + 8 * N])
        # log's table has B[0] = -0.5 at [7] and 128 (invc, logc) pairs with invc in (0.7, 1.45); pow's log table differs
        if d[7] == -0.5 and all(0.69 < d[18 + 2 * i] < 1.46 and abs(d[19 + 2 * i] + math.log(d[18 + 2 * i])) < 1e-9 for i in range(128)):
            return list(vals)
        pos += 8


if __name__ == "__main__":
    t = table()
    for i in range(0, N, 4):
        print("    " + ", ".join("0x%016xull" % v for v in t[i:i + 4]) + ",")
