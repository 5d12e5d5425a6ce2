#!/usr/bin/env python3
"""Static instruction histogram of the iteration loop of a code-specialised body (hipcc -S on a one-kernel translation unit).
usage: isa_histogram.py <body> <Code> <threads> <waves_per_simd>   e.g. ms_m64_body CodeAppendixCM64 64 2"""
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
body, code, threads, wps = sys.argv[1], sys.argv[2], sys.argv[3], sys.argv[4]
with tempfile.TemporaryDirectory() as td:
    src = os.path.join(td, "k.hip")
    open(src, "w").write(f'#include "{ROOT}/ldpc-lib_amd/csrc/ldpc_spec.hpp"\n#include "{ROOT}/ldpc-lib_amd/csrc/code_appendix_c_m64.hpp"\n'
                         f'extern "C" __global__ void __launch_bounds__({threads}, {wps}) k(const ldpc_spec::SpecArgs a) {{ ldpc_spec::{body}<ldpc_spec::{code}>(a); }}\n')
    out = os.path.join(td, "k.s")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "--cuda-device-only", "-S", src, "-o", out],
                          stderr=subprocess.DEVNULL)
    lines = open(out).read().splitlines()
# the iteration loop = every basic block whose label carries a "Loop" annotation (header or "in Loop: Header=..."), wherever the
# compiler placed it relative to the header
c = collections.Counter()
in_loop = False
for l in lines:
    if re.match(r"^\.LBB\d+_\d+:", l):
        in_loop = "Loop" in l
        continue
    if re.match(r"^\.Lfunc_end", l):
        in_loop = False
    if not in_loop:
        continue
    m = re.match(r"^\s+([a-z_0-9]+)\s", l)
    if m and not l.strip().startswith((".", ";")):
        c[m.group(1)] += 1
meta = {k: next((l.split(":")[1].strip() for l in lines if l.strip().startswith(k)), "?") for k in (".vgpr_count", ".sgpr_count", ".vgpr_spill_count", ".private_segment_fixed_size")}
print(f"# {body}<{code}>  launch_bounds({threads},{wps})  {meta}")
print(f"# instructions in the iteration loop: total {sum(c.values())}, VALU {sum(v for k, v in c.items() if k.startswith('v_'))}, "
      f"LDS {sum(v for k, v in c.items() if k.startswith('ds_'))}, SALU {sum(v for k, v in c.items() if k.startswith('s_'))}, "
      f"VMEM {sum(v for k, v in c.items() if k.startswith(('global_', 'buffer_', 'scratch_', 'flat_')))}")
for k, v in c.most_common():
    print(f"{k:28s}{v}")
