"""CPU tests (no GPU): host logic, the C-ABI surface, the C++ compat layer's build, multi-process orchestration."""
import ctypes as C
import os
import re
import socket
import subprocess
import sys

import numpy as np
import pytest

from ldpc_testlib import GOLDEN_DIR, load_base_matrix, ref_lib, relift

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_build_entry_point_and_abi_exports():
    """build() compiles everything (hipcc cross-compiles without a GPU); the .so loads and exports every symbol that
    include/ldpc_hip.h declares.  No compute call is made here."""
    import __graft_entry__
    __graft_entry__.build()
    import ldpc_lib_amd
    lib = ldpc_lib_amd.load_library()
    hdr = open(os.path.join(ROOT, "include", "ldpc_hip.h")).read()
    names = sorted(set(re.findall(r"\b(ldpc_hip_[a-z0-9_]+)\s*\(", hdr)))
    assert len(names) >= 15
    for n in names:
        assert hasattr(lib, n), f"{n} declared in ldpc_hip.h but not exported"
    assert lib.ldpc_hip_abi_version() == 4


def test_generator_state_conversion(tmp_path):
    """include/ldpc/bp_simulation.h: std::mt19937 <-> (624 words, next index), every block position incl. the freshly seeded one
    (tests/cpp/mt_state_roundtrip.cpp; host only)"""
    exe = tmp_path / "mt_rt"
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "cpp", "mt_state_roundtrip.cpp"),
                           "-o", str(exe), "-L", os.path.join(ROOT, "ldpc-lib_amd"), "-lldpc_hip", "-Wl,-rpath," + os.path.join(ROOT, "ldpc-lib_amd")])
    assert subprocess.check_output([str(exe)], text=True).strip() == "ok"


def test_jit_mode_switch():
    """ldpc_hip_set_jit_mode: process-wide default (0 never / 1 inside ldpc_hip_open / 2 background), returns the previous mode"""
    import ldpc_lib_amd
    lib = ldpc_lib_amd.load_library()
    assert lib.ldpc_hip_set_jit_mode(2) == 1          # the C-ABI's default: compile inside ldpc_hip_open
    assert lib.ldpc_hip_set_jit_mode(0) == 2
    assert lib.ldpc_hip_set_jit_mode(7) < 0 and b"ldpc_hip_set_jit_mode" in lib.ldpc_hip_last_error()
    assert lib.ldpc_hip_set_jit_mode(1) == 0


def test_no_gpu_means_loud_failure_not_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import ldpc_lib_amd
    H = np.zeros((2, 4), dtype=np.int16)
    with pytest.raises(ldpc_lib_amd.LdpcHipError):
        ldpc_lib_amd.LdpcHip(ldpc_lib_amd.DEC_MS, H, 64)


def test_product_never_imports_the_oracle():
    """the oracle is test infrastructure: nothing under ldpc-lib_amd/ or include/ may reference it"""
    bad = []
    for base in ("ldpc-lib_amd", "include"):
        for dp, _, fns in os.walk(os.path.join(ROOT, base)):
            for fn in fns:
                if fn.endswith((".py", ".hip", ".hpp", ".h", ".cpp")):
                    txt = open(os.path.join(dp, fn), errors="ignore").read()
                    if re.search(r"liboracle|ldpc_oracle|harness_oracle|oracle/|_ref/", txt):
                        bad.append(os.path.join(dp, fn))
    assert not bad, bad


def test_compat_layer_builds_and_exports_upstream_surface():
    so = os.path.join(ROOT, "ldpc-lib_amd", "libldpc_compat.so")
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "ldpc-lib_amd", "csrc", "compat")])
    syms = subprocess.check_output(["nm", "-DC", so]).decode()
    for want in ("decod_open(int, int, int, int, int)", "decod_init(void*)", "decod_close(DEC_STATE*)",
                 "min_sum_decod_qc_lm(DEC_STATE*, double*, double*, int, int, double)",
                 "sum_prod_decod_qc_lm(DEC_STATE*, double*, double*, int, int)",
                 "lmin_sum_decod_qc_lm(DEC_STATE*, double*, double*, int, int, double, double)",
                 "imin_sum_decod_qc_lm(DEC_STATE*, double*, double*, int, int, double, double, int, int)",
                 "ldpc::bp_simulation(", "ldpc_bp_simulation_exact", "DEC_FULL_NAME"):
        assert want in syms, want


@pytest.mark.skipif(not os.path.isdir("/root/reference"), reason="upstream tree not mounted")
def test_dropin_translation_units_compile_against_upstream_headers():
    """bp_simulation_dropin.cpp defines exactly upstream's bp_simulation symbol (static_assert inside) and
    decoders_compat.cpp works with upstream's DEC_STATE layout."""
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "ldpc-lib_amd", "csrc", "compat"), "check-upstream"])


def test_relift_rule():
    import ldpc_lib_amd
    from ldpc_testlib import load_base_matrix, relift
    H0 = load_base_matrix()
    for M in (1, 7, 64, 126, 512):
        assert np.array_equal(ldpc_lib_amd.relift_base_matrix(H0, M), relift(H0, M))


class FakeSource:
    """Deterministic per-frame records as a pure function of the global frame index (stands in for the GPU decoder)."""
    n, r = 2048, 1024

    def frames(self, first, B):
        import torch
        f = np.arange(first, first + B, dtype=np.int64)
        h = (f * 2654435761 + 12345) % 1000003
        bad = (h % 23) == 0
        info = np.where(bad, (1 << 30) | (h % 37), 0).astype(np.int32)
        iters = np.where(bad & (h % 3 == 0), -50, 1 + h % 20).astype(np.int32)
        return torch.from_numpy(info), torch.from_numpy(iters)

    def close(self):
        pass


def _brute(n_fe, n_exp, ref):
    src = FakeSource()
    nse = nde = nue = exp = 0
    while nde < n_fe and exp <= n_exp:
        info, it = src.frames(exp, 1)
        exp += 1
        if int(info[0]) & (1 << 30):
            nse += int(info[0]) & ((1 << 30) - 1); nde += 1; nue += int(it[0]) >= 0
            if nde >= 10 and nde / exp > 2.5 * ref:
                break
    return nse, nde, nue, exp


@pytest.mark.parametrize("n_fe,n_exp,ref,batch", [(10**9, 999, 1.0, 64), (25, 10**6, 1.0, 100), (10**9, 5000, 0.001, 257), (3, 50, 1.0, 1000)])
def test_batched_host_harness_equals_frame_by_frame_loop(n_fe, n_exp, ref, batch):
    import ldpc_lib_amd
    _, _, st = ldpc_lib_amd.bp_simulation(None, 64, 50, n_fe, n_exp, 2.0, ref, batch=batch, source=FakeSource(), return_state=True)
    assert (st["nse"], st["nde"], st["nue"], st["experiment"]) == _brute(n_fe, n_exp, ref)


_WORKER = r"""
import os, sys, json
sys.path.insert(0, {root!r}); sys.path.insert(0, os.path.join({root!r}, "tests"))
import torch.distributed as dist
import ldpc_lib_amd
from test_host_cpu import FakeSource
dist.init_process_group("gloo", rank=int(os.environ["RANK"]), world_size=int(os.environ["WORLD_SIZE"]))
out = []
for n_fe, n_exp, ref, batch in [(10**9, 999, 1.0, 64), (25, 10**6, 1.0, 100), (10**9, 5000, 0.001, 257), (3, 50, 1.0, 1000)]:
    _, _, st = ldpc_lib_amd.bp_simulation(None, 64, 50, n_fe, n_exp, 2.0, ref, batch=batch, source=FakeSource(), return_state=True)
    out.append([st["nse"], st["nde"], st["nue"], st["experiment"]])
print("RESULT", dist.get_rank(), json.dumps(out))
dist.destroy_process_group()
"""


def test_two_rank_gloo_run_matches_single_process():
    """N > 1 path on CPU: world_size 2 over gloo; frames sharded by global index, per-frame records all-gathered,
    every rank replays the stopping rule -> both ranks report the single-process result."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, "-c", _WORKER.format(root=ROOT)], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=300)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    import json
    want = [list(_brute(*c[:3])) for c in [(10**9, 999, 1.0), (25, 10**6, 1.0), (10**9, 5000, 0.001), (3, 50, 1.0)]]
    for o in outs:
        line = [ln for ln in o.splitlines() if ln.startswith("RESULT")][0]
        assert json.loads(line.split(" ", 2)[2]) == want


def _fake_mt_source():
    """The exact-replay source without a GPU: the `generator` is a frame counter (a sequential stream every rank must advance
    identically, snapshot / restore-and-skip included), the records are FakeSource's function of the stream position."""
    import ldpc_lib_amd

    class FakeMtSource(ldpc_lib_amd.MtFrameSource):
        n, r = 2048, 1024

        def __init__(self):
            self.pos = 0

        def snapshot(self):
            return self.pos

        def restore_and_skip(self, snap, frames):
            self.pos = snap + frames

        def round(self, total, lo, hi):
            info, iters = FakeSource().frames(self.pos + lo, hi - lo)
            self.pos += total
            return info, iters

        def close(self):
            pass

    return FakeMtSource()


@pytest.mark.parametrize("n_fe,n_exp,ref,batch", [(10**9, 999, 1.0, 64), (25, 10**6, 1.0, 100), (10**9, 5000, 0.001, 257), (3, 50, 1.0, 1000)])
def test_exact_replay_branch_of_the_host_harness(n_fe, n_exp, ref, batch):
    """sequential-generator bookkeeping: same counters as the frame-by-frame loop, and the generator ends exactly `experiment`
    frames on, also when the run stops inside a round"""
    import ldpc_lib_amd
    _, _, st = ldpc_lib_amd.bp_simulation(None, 64, 50, n_fe, n_exp, 2.0, ref, batch=batch, source=_fake_mt_source(), return_state=True)
    assert (st["nse"], st["nde"], st["nue"], st["experiment"]) == _brute(n_fe, n_exp, ref)
    assert st["generator"] == st["experiment"]


_EXACT_WORKER = r"""
import os, sys, json
sys.path.insert(0, {root!r}); sys.path.insert(0, os.path.join({root!r}, "tests"))
import torch.distributed as dist
import ldpc_lib_amd
from test_host_cpu import _fake_mt_source
dist.init_process_group("gloo", rank=int(os.environ["RANK"]), world_size=int(os.environ["WORLD_SIZE"]))
out = []
for n_fe, n_exp, ref, batch in [(10**9, 999, 1.0, 64), (25, 10**6, 1.0, 100), (10**9, 5000, 0.001, 257), (3, 50, 1.0, 1000)]:
    _, _, st = ldpc_lib_amd.bp_simulation(None, 64, 50, n_fe, n_exp, 2.0, ref, batch=batch, source=_fake_mt_source(), return_state=True)
    out.append([st["nse"], st["nde"], st["nue"], st["experiment"], st["generator"]])
print("RESULT", dist.get_rank(), json.dumps(out))
dist.destroy_process_group()
"""


def test_two_rank_gloo_exact_replay_matches_single_process():
    """N > 1 exact replay on CPU: every rank advances the same sequential generator by the whole round and takes its slice."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, "-c", _EXACT_WORKER.format(root=ROOT)], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=300)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    import json
    want = [list(_brute(*c[:3])) for c in [(10**9, 999, 1.0), (25, 10**6, 1.0), (10**9, 5000, 0.001), (3, 50, 1.0)]]
    want = [w + [w[3]] for w in want]
    for o in outs:
        line = [ln for ln in o.splitlines() if ln.startswith("RESULT")][0]
        assert json.loads(line.split(" ", 2)[2]) == want


def test_bench_starts_its_own_ranks_and_fails_loudly_without_a_gpu():
    """`python bench.py --gpus 2` with no launcher: the parent starts one fresh process per rank itself (it used to exit
    with "launch with torch.distributed.run").  Without a GPU every rank must refuse loudly -- there is no CPU fallback to time -- and
    the parent must report it with a non-zero exit code instead of hanging.  (The run itself is tests/test_gpu_chain.py's.)"""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present: test_gpu_chain.py runs the real thing")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--steps", "2", "--warmup", "1"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode != 0
    assert "rank 0/2: no GPU" in p.stderr and "rank 1/2: no GPU" in p.stderr, p.stderr   # both ranks were started, with the right world
    assert not [ln for ln in p.stdout.splitlines() if ln.startswith("{")]              # and no result line was made up


def test_multi_gpu_test_programs_build(tmp_path):
    """tests/cpp/multi_driver.c and the host-memory stand-in for librccl (tests/cpp/rccl_stub.cpp) that the GPU suite uses to drive the
    N > 1 communicator code on a one-GPU box: they must build here, and the stand-in must export what csrc/ldpc_multi.hpp resolves."""
    import ctypes
    hipinc = ["-D__HIP_PLATFORM_AMD__", "-I", "/opt/rocm/include"]
    stub = tmp_path / "librccl_stub.so"
    subprocess.check_call(["g++", "-O1", "-Wall", "-fPIC", "-shared", *hipinc, os.path.join(ROOT, "tests", "cpp", "rccl_stub.cpp"), "-o", str(stub),
                           "-L", "/opt/rocm/lib", "-lamdhip64", "-Wl,-rpath,/opt/rocm/lib"])
    subprocess.check_call(["gcc", "-O1", "-Wall", "-Werror", *hipinc, "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "cpp", "multi_driver.c"),
                           "-o", str(tmp_path / "multi_driver"), "-L", os.path.join(ROOT, "ldpc-lib_amd"), "-lldpc_hip", "-L", "/opt/rocm/lib", "-lamdhip64",
                           "-Wl,-rpath," + os.path.join(ROOT, "ldpc-lib_amd"), "-Wl,-rpath,/opt/rocm/lib"])
    lib = ctypes.CDLL(str(stub))
    src = open(os.path.join(ROOT, "ldpc-lib_amd", "csrc", "ldpc_multi.hpp")).read()
    wanted = re.findall(r"LDPC_RCCL_SYM\((\w+)\)", src.split("#define LDPC_RCCL_SYM")[1].split("#undef")[0])
    assert len(wanted) >= 6
    for w in wanted:
        assert hasattr(lib, "nccl" + w), w


def test_c_example_builds_against_the_header(tmp_path):
    """examples/simulate.c, simulate_multi.c and exact_replay.c are plain C: the header must be C-clean and the library must link from gcc."""
    for name in ("simulate", "simulate_multi", "exact_replay"):
        exe = tmp_path / name
        subprocess.check_call(["gcc", "-O1", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "examples", name + ".c"),
                               "-o", str(exe), "-L", os.path.join(ROOT, "ldpc-lib_amd"), "-lldpc_hip",
                               "-Wl,-rpath," + os.path.join(ROOT, "ldpc-lib_amd")])
        assert exe.exists()


# ---- jsonx reader / writer and the `ldpc_sim` driver -------------------------------------------------------
def _ldpc_sim():
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    subprocess.check_call(["make", "-s", "-C", os.path.join(root, "ldpc-lib_amd", "csrc", "compat")])
    exe = os.path.join(root, "ldpc-lib_amd", "ldpc_sim")
    assert os.path.exists(exe)
    return exe


def test_jsonx_reader_follows_upstreams_grammar(tmp_path):
    """Every construct upstream's settings.cpp:178-345 reads: records, arrays, matrices, sparse matrices, strings, numbers kept
    as text, '/'-comments, '@' file references (relative to the including file), `array @file`, first-key-wins, and select()
    falling back to `defaults` records."""
    import subprocess
    exe = _ldpc_sim()
    (tmp_path / "sub").mkdir()
    (tmp_path / "sub" / "consts.jsonx").write_text('{ alpha = 0.8   / a single slash starts a comment\n beta = 4e-1 nested = { deep = "yes" } }')
    (tmp_path / "sub" / "codes.jsonx").write_text('{ id = 0 m = matrix (2 3) { 0 -1 7  5 -1 -1 } }\n// second value of the sequence\n{ id = 1 }')
    (tmp_path / "main.jsonx").write_text("""// scenario
{
    defaults = @"sub/consts.jsonx"
    snrs = array { 1.7 }
    snrs = array { 2.7 }          // ignored: the first value of a key wins (std::map::insert)
    name = "two words"
    results = array @"sub/codes.jsonx"
    sp = sparse matrix (2 2) { 0 1 5   1 0 -3 }
    settings = { defaults = { inner = 3 } x = 1e-1 list = array { array { 1 2 } array { } } }
}""")
    def get(path):
        return subprocess.check_output([exe, "jsonx-get", str(tmp_path / "main.jsonx"), path], text=True).strip()
    assert get("snrs") == "array { 1.7 }"
    assert get("name") == "two words"
    assert get("alpha") == "0.8" and get("beta") == "4e-1" and get("nested/deep") == "yes"     # through `defaults`, read from a file
    assert get("settings/inner") == "3" and get("settings/x") == "1e-1"
    assert get("results/0/m").split() == "matrix (2 3) { 0 -1 7 5 -1 -1 }".split()
    assert get("results/1/id") == "1"
    assert get("sp").split()[:4] == ["matrix", "(2", "2)", "{"]
    assert get("settings/list") .split() == "array { array { 1 2 } array { } }".split()
    assert subprocess.run([exe, "jsonx-get", str(tmp_path / "main.jsonx"), "nope"], capture_output=True).returncode == 1
    # canonical form is a fixed point, and it is readable again
    subprocess.check_call([exe, "jsonx", str(tmp_path / "main.jsonx"), str(tmp_path / "a.jsonx")])
    subprocess.check_call([exe, "jsonx", str(tmp_path / "a.jsonx"), str(tmp_path / "b.jsonx")])
    assert (tmp_path / "a.jsonx").read_text() == (tmp_path / "b.jsonx").read_text()
    (tmp_path / "bad.jsonx").write_text("{ a = array { 1 2 ")
    assert subprocess.run([exe, "jsonx", str(tmp_path / "bad.jsonx"), str(tmp_path / "c.jsonx")], capture_output=True).returncode == 1


@pytest.mark.skipif(not os.path.isdir("/root/reference/files"), reason="upstream tree not mounted")
def test_jsonx_reader_takes_every_file_upstream_ships(tmp_path):
    import glob
    import subprocess
    exe = _ldpc_sim()
    files = sorted(glob.glob("/root/reference/files/*.jsonx"))
    assert len(files) >= 10
    for f in files:
        subprocess.check_call([exe, "jsonx", f, str(tmp_path / "a.jsonx")])
        subprocess.check_call([exe, "jsonx", str(tmp_path / "a.jsonx"), str(tmp_path / "b.jsonx")])
        assert (tmp_path / "a.jsonx").read_text() == (tmp_path / "b.jsonx").read_text(), f
    def get(f, path):
        return subprocess.check_output([exe, "jsonx-get", "/root/reference/files/" + f, path], text=True).strip()
    assert get("input32_16.jsonx", "decoder_type") == "7" and get("input32_16.jsonx", "permutation_block") == "128"
    assert get("resultq.jsonx", "results/0/_lifting") == "8" and get("resultq.jsonx", "settings/error_minimization/name") == "FER"


def test_example_scenario_is_well_formed():
    import subprocess
    exe = _ldpc_sim()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    ex = os.path.join(root, "examples", "simulation_appendix_c.jsonx")
    def get(path):
        return subprocess.check_output([exe, "jsonx-get", ex, path], text=True).strip()
    assert get("settings/num_codewords") == "2000" and get("results/2/_lifting") == "512" and get("results/3/_SNRs") == "array { 2.0 2.5 }"
    from ldpc_testlib import load_base_matrix
    cells = [int(x) for x in get("results/0/code").replace("matrix (16 32) {", "").replace("}", "").split()]
    assert np.array_equal(np.array(cells).reshape(16, 32), load_base_matrix())


def test_interleaver_maps_equal_upstreams():
    """ldpc/interleaver.h (through the C-ABI, host only) against the maps recorded from the compiled upstream interleaver
    (oracle/make_perm_goldens.py): identity, random, deterministic (all four halfmlog values, block-column counts that are and
    are not multiples of halfmlog), block random with a short tail, interleaved random."""
    import ldpc_lib_amd  # noqa: F401
    from ldpc_lib_amd.binding import LdpcHipError, build_interleaver
    g = np.load(os.path.join(GOLDEN_DIR, "interleavers.npz"))
    keys = sorted(k[:-4] for k in g.files if k.endswith("_cfg"))
    assert len(keys) >= 200
    modes = set()
    for k in keys:
        M, h, mode, bs, st = (int(x) for x in g[k + "_cfg"])
        d, i = build_interleaver(g[k + "_H"], M, mode, h, bs, st)
        assert np.array_equal(d, g[k + "_direct"]) and np.array_equal(i, g[k + "_inverse"]), (k, M, h, mode, bs, st)
        assert np.array_equal(d[i], np.arange(d.size))          # the two directions undo each other
        modes.add((mode, h))
    assert {(m, h) for m in range(5) for h in (1, 2, 3, 4)} <= modes
    H = g[keys[0] + "_H"]
    with pytest.raises(LdpcHipError):
        build_interleaver(H, 8, 4, 1, 0, 5)      # step that does not divide N: upstream leaves positions unwritten -> rejected
    with pytest.raises(LdpcHipError):
        build_interleaver(H, 8, 7, 1)            # unknown permutation type


@pytest.mark.skipif(ref_lib() is None, reason="oracle/_ref not built (needs the upstream tree)")
def test_interleaver_maps_against_the_compiled_reference_beyond_the_fixture():
    import ctypes as C
    import ldpc_lib_amd  # noqa: F401
    from ldpc_lib_amd.binding import build_interleaver
    H = np.ascontiguousarray(relift(load_base_matrix(), 64), dtype=np.int16)
    for h, mode, bs, st in ((1, 1, 0, 0), (2, 2, 0, 0), (3, 2, 0, 0), (4, 2, 0, 0), (1, 3, 128, 0), (1, 3, 300, 0), (1, 4, 0, 1), (2, 4, 0, 16)):
        d = np.zeros(2048, dtype=np.int32)
        i = np.zeros(2048, dtype=np.int32)
        assert ref_lib().ref_perm_maps(16, 32, 64, 1 << (2 * h), h, mode, bs, st, H.ctypes.data_as(C.c_void_p), d.ctypes.data_as(C.c_void_p),
                                       i.ctypes.data_as(C.c_void_p)) == 0
        dd, ii = build_interleaver(H, 64, mode, h, bs, st)
        assert np.array_equal(d, dd) and np.array_equal(i, ii), (h, mode, bs, st)


def test_host_headers_under_address_and_ub_sanitizers(tmp_path):
    """jsonx reader/writer and interleaver maps with -fsanitize=address,undefined: malformed, mutated and hostile inputs end in
    jsonx::Error / a refusal, never in a crash, an out-of-bounds access or undefined behaviour (tests/cpp/host_sanitize.cpp)."""
    import subprocess
    exe = str(tmp_path / "host_sanitize")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cpp", "host_sanitize.cpp"), "-o", exe])
    out = subprocess.run([exe], capture_output=True, text=True)
    assert out.returncode == 0 and out.stdout.strip().endswith("ok"), out.stdout[-2000:] + out.stderr[-2000:]


def test_encoder_returns_the_unique_systematic_codeword():
    """include/ldpc/encoder.h through the C-ABI (host only): for the example code upstream's search produced, at several
    liftings, the result is systematic, satisfies every parity check (checked with the independent numpy syndrome), is linear
    in the information bits, and maps zero to zero.  A parity part the dual-diagonal encoder cannot solve is refused."""
    import ldpc_lib_amd  # noqa: F401
    from ldpc_lib_amd.binding import LdpcHipError, encode
    from ldpc_testlib import syndrome_np
    rng = np.random.RandomState(11)
    for M in (1, 5, 64, 67, 126, 512):
        H = relift(load_base_matrix(), M)
        K = 16 * M
        a, b = rng.randint(0, 2, K).astype(np.uint8), rng.randint(0, 2, K).astype(np.uint8)
        ca, cb, cab = encode(H, M, a), encode(H, M, b), encode(H, M, a ^ b)
        assert np.array_equal(ca[K:], a) and np.array_equal(cb[K:], b)
        assert not syndrome_np(H, M, np.stack([ca, cb, cab])).any()
        assert np.array_equal(ca ^ cb, cab)
        assert not encode(H, M, np.zeros(K, dtype=np.uint8)).any()
    from ldpc_testlib import multi_block_code
    for M, blocks in ((64, (4, 4)), (7, (3, 5, 4)), (126, (5, 3))):      # several dual-diagonal blocks (bp_simulation.cpp:142-191)
        Hb = multi_block_code(np.random.RandomState(77 + M), M, blocks)
        K = (Hb.shape[1] - sum(blocks)) * M
        a, b = rng.randint(0, 2, K).astype(np.uint8), rng.randint(0, 2, K).astype(np.uint8)
        ca, cb, cab = encode(Hb, M, a), encode(Hb, M, b), encode(Hb, M, a ^ b)
        assert np.array_equal(ca[sum(blocks) * M:], a) and not syndrome_np(Hb, M, np.stack([ca, cb, cab])).any() and np.array_equal(ca ^ cb, cab)
    bad = relift(load_base_matrix(), 8).copy()
    bad[:, 15] = -1                       # no special parity column any more
    bad[15, 15] = 0
    with pytest.raises(LdpcHipError):
        encode(bad, 8, np.ones(16 * 8, dtype=np.uint8))


def test_device_exp_algorithm_equals_libm_exp_bit_for_bit(tmp_path):
    """ldpc_spec::exp_glibc (the one transcendental of the SP / ASP / TDMP decoders) is glibc's exp() algorithm with a table
    computed by tools/gen_exp_table.py.  A C transcription of exactly that code -- same constants, same table, same fma
    placement -- must return libm's exp() bit for bit on this host (x86-64 with FMA: glibc's run-time selected FMA build), over
    the decoders' whole input range; and the header's table must be the generator's."""
    import subprocess
    flags = open("/proc/cpuinfo").read()
    if " fma" not in flags:
        pytest.skip("host without FMA: glibc selects its non-FMA exp build there")
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import gen_exp_table
    tab = gen_exp_table.table()
    hdr = open(os.path.join(ROOT, "ldpc-lib_amd", "csrc", "ldpc_spec.hpp")).read()
    body = hdr[hdr.index("kExpTab[256] = {") + len("kExpTab[256] = {"):]
    body = body[:body.index("};")]
    assert [int(x.strip().rstrip("ull"), 16) for x in body.replace("\n", " ").split(",") if x.strip()] == tab
    fn = hdr[hdr.index("__device__ __forceinline__ double exp_glibc_t(double x, Tab T) {"):]
    fn = fn[:fn.index("\n}\n") + 3]
    c_fn = (fn.replace("__device__ __forceinline__ double exp_glibc_t(double x, Tab T)", "static double exp_glibc(double x)").replace("__fma_rn", "fma")
              .replace("const ulonglong2 pair = *reinterpret_cast<const ulonglong2 *>(&T[idx]);", "const struct { unsigned long long x, y; } pair = {kExpTab[idx], kExpTab[idx + 1]};")
              .replace("(unsigned long long)__double_as_longlong(kd)", "asu(kd)")
              .replace("__longlong_as_double((long long)pair.x)", "asd(pair.x)")
              .replace("__longlong_as_double((long long)sbits)", "asd(sbits)"))
    src = ("#include <math.h>\n#include <stdint.h>\n#include <stdio.h>\n#include <string.h>\n"
           "static unsigned long long asu(double x){unsigned long long u;memcpy(&u,&x,8);return u;}\n"
           "static double asd(unsigned long long u){double x;memcpy(&x,&u,8);return x;}\n"
           "static const unsigned long long kExpTab[256] = {" + ",".join("0x%xull" % v for v in tab) + "};\n" + c_fn +
           "int main(){unsigned long long s=88172645463325252ull;long bad=0,n=4000000;\n"
           " for(long i=0;i<n;i++){s^=s<<13;s^=s>>7;s^=s<<17;double x=((double)(s>>11)/9007199254740992.0)*42.0-21.0;\n"
           "  if(i<2000) x=(i-1000)*1e-19; if(asu(exp_glibc(x))!=asu(exp(x))) bad++;}\n"
           " printf(\"%ld\\n\",bad);return bad!=0;}\n")
    (tmp_path / "e.c").write_text(src)
    subprocess.check_call(["gcc", "-O2", "-ffp-contract=off", "-mfma", str(tmp_path / "e.c"), "-o", str(tmp_path / "e"), "-lm"])
    assert subprocess.check_output([str(tmp_path / "e")], text=True).strip() == "0"


def test_device_log_algorithm_equals_libm_log_bit_for_bit(tmp_path):
    """ldpc_spec::log_glibc (Gallager BP's log) is glibc's log() algorithm with the data tools/gen_log_table.py reads from this
    host's libm.  A C transcription of exactly the header's code -- same data, same fma placement -- must return libm's log() bit
    for bit on this host (x86-64 with FMA: glibc's run-time selected FMA build): (0,1), (1,inf), the near-1 path, subnormals,
    and the special values 0, 1, inf."""
    import subprocess
    if " fma" not in open("/proc/cpuinfo").read():
        pytest.skip("host without FMA: glibc selects its non-FMA log build there")
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import gen_log_table
    tab = gen_log_table.table()
    hdr = open(os.path.join(ROOT, "ldpc-lib_amd", "csrc", "ldpc_spec.hpp")).read()
    body = hdr[hdr.index("kLogData[274] = {") + len("kLogData[274] = {"):]
    body = body[:body.index("};")]
    assert [int(x.strip().rstrip("ull"), 16) for x in body.replace("\n", " ").split(",") if x.strip()] == tab
    fn = hdr[hdr.index("__device__ __forceinline__ double log_glibc_t(double x, Tab T) {"):]
    fn = fn[:fn.index("\n}\n") + 3]
    c_fn = (fn.replace("__device__ __forceinline__ double log_glibc_t(double x, Tab T)", "static double log_glibc(double x)").replace("__fma_rn", "fma")
              .replace("    auto D = [&](int i) { return __longlong_as_double((long long)kLogData[i]); };\n", "")
              .replace("const ulonglong2 cpair = *reinterpret_cast<const ulonglong2 *>(&T[18 + 2 * i]);", "const struct { unsigned long long x, y; } cpair = {kLogData[18 + 2 * i], kLogData[19 + 2 * i]};")
              .replace("__longlong_as_double((long long)cpair.x)", "asd(cpair.x)").replace("__longlong_as_double((long long)cpair.y)", "asd(cpair.y)")
              .replace("(unsigned long long)__double_as_longlong(", "asu(").replace("__longlong_as_double((long long)iz)", "asd(iz)")
              .replace("__longlong_as_double(0x7ff0000000000000ll)", "asd(0x7ff0000000000000ull)")
              .replace("__longlong_as_double(0x7ff8000000000000ll)", "asd(0x7ff8000000000000ull)")
              )
    src = ("#include <math.h>\n#include <stdint.h>\n#include <stdio.h>\n#include <string.h>\n"
           "static unsigned long long asu(double x){unsigned long long u;memcpy(&u,&x,8);return u;}\n"
           "static double asd(unsigned long long u){double x;memcpy(&x,&u,8);return x;}\n"
           "static const unsigned long long kLogData[274] = {" + ",".join("0x%xull" % v for v in tab) + "};\n"
           "typedef unsigned u32;\n#define D(i) asd(kLogData[i])\n" + c_fn +
           "int main(){unsigned long long s=88172645463325252ull;long bad=0,n=6000000;\n"
           " double sp[5]={0.0,1.0,1.0/0.0,0x1p-1074,0x1.fffffffffffffp1023};\n"
           " for(int i=0;i<5;i++) if(asu(log_glibc(sp[i]))!=asu(log(sp[i]))) bad++;\n"
           " if(!isnan(log_glibc(-1.0))||!isnan(log_glibc(0.0/0.0))) bad++;\n"
           " for(long i=0;i<n;i++){s^=s<<13;s^=s>>7;s^=s<<17;double u=((double)(s>>11)/9007199254740992.0),x;int m=i%4;\n"
           "  if(m==0) x=u; else if(m==1) x=1.0/(u+1e-300); else if(m==2) x=0.9375+u*0.13; else x=ldexp(u,-(int)(s%1070));\n"
           "  if(asu(log_glibc(x))!=asu(log(x))) bad++;}\n"
           " printf(\"%ld\\n\",bad);return bad!=0;}\n")
    (tmp_path / "l.c").write_text(src)
    subprocess.check_call(["gcc", "-O2", "-ffp-contract=off", "-mfma", str(tmp_path / "l.c"), "-o", str(tmp_path / "l"), "-lm"])
    assert subprocess.check_output([str(tmp_path / "l")], text=True).strip() == "0"


def test_bench_children_do_not_inherit_a_launchers_rendezvous(monkeypatch):
    """bench.py starts jobs of its own (the ranks of `--gpus N`, the abi_multi and exact_replay_ranks legs) from inside a rank that
    torch.distributed.run started: with the launcher's TORCHELASTIC_USE_AGENT_STORE=True in the child's environment its rank 0 would
    not host the rendezvous store and the child job would hang (seen on the GPU box, round 3)."""
    sys.path.insert(0, ROOT)
    import bench
    for k, v in {"RANK": "3", "LOCAL_RANK": "3", "WORLD_SIZE": "8", "MASTER_ADDR": "10.0.0.1", "MASTER_PORT": "29500", "GROUP_RANK": "0",
                 "ROLE_RANK": "3", "ROLE_NAME": "default", "LOCAL_WORLD_SIZE": "8", "ROLE_WORLD_SIZE": "8", "GROUP_WORLD_SIZE": "1",
                 "TORCHELASTIC_USE_AGENT_STORE": "True", "TORCHELASTIC_RUN_ID": "x", "TORCHELASTIC_RESTART_COUNT": "0",
                 "TORCHELASTIC_MAX_RESTARTS": "0", "LDPC_HIP_CACHE_DIR": "/tmp/keep"}.items():
        monkeypatch.setenv(k, v)
    env = bench.clean_env()
    assert not [k for k in env if k.startswith("TORCHELASTIC_") or k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT",
                                                                       "GROUP_RANK", "ROLE_RANK", "LOCAL_WORLD_SIZE", "GROUP_WORLD_SIZE")]
    assert env["LDPC_HIP_CACHE_DIR"] == "/tmp/keep" and env["HSA_ENABLE_IPC_MODE_LEGACY"] == "0" and "PATH" in env


_RANK_PROTOCOL_WORKER = r"""
import os, sys, json
sys.path.insert(0, {root!r}); sys.path.insert(0, os.path.join({root!r}, "tests"))
import numpy as np, torch, torch.distributed as dist
import ldpc_lib_amd
dist.init_process_group("gloo", rank=int(os.environ["RANK"]), world_size=int(os.environ["WORLD_SIZE"]))
rank, world = dist.get_rank(), dist.get_world_size()

class FakeDec:
    # the C-ABI calls MtFrameSource.round() makes, on a tape whose 'generator state' is the number of frames drawn so far; frame f's
    # record is (f, 1).  script[k] says how round k goes: ("ok",), ("uncovered", r): rank r's window does not cover its rows,
    # ("nobody",): no rank finds the end state, ("short", d): the round completes only total - d frames
    device = None
    def __init__(self, script):
        self.pos, self.script, self.k, self.log = 0, script, 0, []
    def mt_shard_begin(self, snr, total, rank, n, modulation=0, punctured_blocks=0):
        self.cur = (total, rank, n, self.script[self.k]); self.k += 1
        return (total * (rank + 1) // n - total * rank // n) * 7
    def mt_shard_emit(self, counts):
        total, rank, n, how = self.cur
        assert len(counts) == n and sum(counts) == total * 7
        fdone = total - how[1] if how[0] == "short" else total
        st = np.zeros(624, dtype=np.uint32); st[0] = self.pos + fdone
        found = rank == n - 1 and how[0] != "nobody"
        covered = not (how[0] == "uncovered" and how[1] == rank)
        return found, covered, (st if found else np.zeros(624, dtype=np.uint32)), fdone
    def mt_shard_commit(self, state, fdone, maxit, rows, alpha=0.8):
        total, rank, n, how = self.cur
        lo = total * rank // n
        ids = np.arange(self.pos + lo, self.pos + lo + rows, dtype=np.int32)
        self.pos = int(state[0]); self.log.append("commit")
        return ids, np.ones(rows, dtype=np.int32)
    def mt_shard_abandon(self):
        self.log.append("abandon")
    def mt_frames(self, snr, maxit, B, modulation=0, punctured_blocks=0, alpha=0.8, lo=None, hi=None):
        lo = 0 if lo is None else lo; hi = B if hi is None else hi
        ids = np.arange(self.pos + lo, self.pos + hi, dtype=np.int32)
        self.pos += B; self.log.append("frames")
        return ids, np.ones(hi - lo, dtype=np.int32)

script = [("ok",), ("uncovered", world - 1), ("ok",), ("short", 5), ("nobody",), ("ok",)]
src = ldpc_lib_amd.MtFrameSource.__new__(ldpc_lib_amd.MtFrameSource)
src.dec, src.n, src.r, src.args = FakeDec(script), 2048, 1024, (2.0, 0, 0, 50, 0.8)
src.share_tape, src.shared_rounds, src.fallback_rounds = True, 0, 0
seen, first = [], 0
for total in (30, 31, 64, 50, 33, 9 * world):
    lo, hi = total * rank // world, total * (rank + 1) // world
    info, its = src.round(total, lo, hi)
    rows = torch.tensor([len(info)]); allrows = [torch.zeros_like(rows) for _ in range(world)]
    dist.all_gather(allrows, rows)
    pad = torch.full((total,), -1, dtype=torch.int32); pad[:len(info)] = info.cpu()
    got = [torch.zeros_like(pad) for _ in range(world)]
    dist.all_gather(got, pad)
    ids = np.concatenate([g.numpy()[:int(a)] for g, a in zip(got, allrows)])
    assert np.array_equal(ids, np.arange(first, first + total)), (total, ids[:8], first)
    first += total
    assert src.dec.pos == first
print("RESULT", rank, json.dumps([src.shared_rounds, src.fallback_rounds, src.dec.log]))
dist.destroy_process_group()
"""


@pytest.mark.parametrize("world", [2, 3])
def test_rank_protocol_of_the_sharded_exact_replay_on_a_scripted_tape(world):
    """host.MtFrameSource.round() over gloo against a stand-in for the C-ABI calls: a round that goes through, one whose last rank is not
    covered (every rank must abandon and run the whole tape), a short round (the remainder the plain way, on every rank), one in which
    nobody finds the end state (fallback again).  Every rank must hand back exactly its slice of consecutive frames and end with the
    same generator position."""
    import json
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, "-c", _RANK_PROTOCOL_WORKER.format(root=ROOT)], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=300)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    for o in outs:
        shared, fallback, log = json.loads([ln for ln in o.splitlines() if ln.startswith("RESULT")][0].split(" ", 2)[2])
        assert (shared, fallback) == (4, 2)
        assert log == ["commit", "abandon", "frames", "commit", "commit", "frames", "abandon", "frames", "commit"]
