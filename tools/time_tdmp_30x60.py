import sys, time, os
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, torch
import ldpc_lib_amd
from ldpc_testlib import random_qc_code, TASP_DEC
rng = np.random.RandomState(67)
H = random_qc_code(rng, 30, 60, 67, [2, 3, 3, 16, 2, 3])
for mode in ("resident", "global"):
    if mode == "global": os.environ["LDPC_HIP_FORCE_GLOBAL"] = "1"
    t0 = time.perf_counter()
    dec = ldpc_lib_amd.LdpcHip(TASP_DEC, H, 67)
    t_open = time.perf_counter() - t0
    llr = dec.awgn_llr(1.4, 1, 0, 8192)
    ts = []
    for r in range(4):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        hard, iters, _ = dec.decode(llr, 15)
        torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    print(mode, dec.kernel_name, "open %.2f s" % t_open, "decode %.3f ms" % (min(ts[1:]) * 1e3), "mean it %.2f" % float(iters.abs().double().mean()), int(iters.sum()))
    dec.close()
