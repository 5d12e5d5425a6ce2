"""ctypes binding of the C-ABI in include/ldpc_hip.h (what cgo / JNI / any FFI would bind the same way).

Device buffers are torch CUDA tensors (torch = allocator + streams only); host entry points take numpy arrays laid
out exactly like the upstream per-frame arrays.
"""
import ctypes as C
import os
import subprocess

import numpy as np

# decoders.h:16-28 enum DEC_ID
DEC_BP, DEC_SP, DEC_ASP, DEC_MS, DEC_IMS, DEC_TASP, DEC_LMS = 0, 1, 2, 3, 4, 7, 8

_PKG_DIR = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_PKG_DIR)
_LIB_NAME = "libldpc_hip.so"
_lib = None


class LdpcHipError(RuntimeError):
    pass


def library_path():
    return os.path.join(_PKG_DIR, _LIB_NAME)


def build_library(force=False, verbose=False, jobs=None):
    """Compile csrc/ldpc_hip.hip (host side, table-driven / shape-unlimited / front-end / generator kernels) and csrc/aot/*.hip (the
    ahead-of-time code-specialised instances) for gfx950 -- one hipcc per translation unit, in parallel -- and link them into
    ldpc-lib_amd/libldpc_hip.so (in-tree, so it travels with the repo snapshot).
    -ffp-contract=off is part of the numerics contract: `y + s*alpha` must stay two roundings (decoders.cpp:4682)."""
    from concurrent.futures import ThreadPoolExecutor
    src_dir = os.path.join(_PKG_DIR, "csrc")
    aot_dir = os.path.join(src_dir, "aot")
    srcs = [os.path.join(src_dir, "ldpc_hip.hip")] + sorted(os.path.join(aot_dir, f) for f in os.listdir(aot_dir) if f.endswith(".hip"))
    headers = [os.path.join(src_dir, f) for f in os.listdir(src_dir) if f.endswith(".hpp")]
    headers += [os.path.join(_ROOT, "include", "ldpc_hip.h"), os.path.join(_ROOT, "include", "ldpc", "interleaver.h"),
                os.path.join(_ROOT, "include", "ldpc", "encoder.h")]
    aot_headers = [os.path.join(src_dir, f) for f in ("ldpc_aot.hpp", "ldpc_spec.hpp", "code_appendix_c_m64.hpp")]
    obj_dir = os.path.join(_PKG_DIR, "build")
    os.makedirs(obj_dir, exist_ok=True)
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-I", os.path.join(_ROOT, "include")]

    def newer(target, deps):
        return os.path.exists(target) and all(os.path.getmtime(d) <= os.path.getmtime(target) for d in deps)

    todo, objs = [], []
    for src in srcs:
        obj = os.path.join(obj_dir, os.path.basename(src)[:-4] + ".o")
        objs.append(obj)
        deps = [src] + (aot_headers if os.path.dirname(src) == aot_dir else headers)
        if force or not newer(obj, deps):
            todo.append([hipcc, *flags, "-c", src, "-o", obj])
    out = library_path()
    if not todo and newer(out, objs):
        return out

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)

    with ThreadPoolExecutor(max_workers=jobs or min(8, os.cpu_count() or 1)) as ex:
        list(ex.map(run, todo))
    run([hipcc, "--offload-arch=gfx950", "-fPIC", "-shared", *objs, "-o", out])
    return out


def load_library():
    """Load libldpc_hip.so.  Raises LdpcHipError when it is missing -- there is no other compute path."""
    global _lib
    if _lib is not None:
        return _lib
    # torch ships its own ROCm runtime (libamdhip64).  Load torch FIRST so that this process has exactly one HIP
    # runtime and our library binds to the same one that owns torch's allocations and streams; loading ours first
    # pulls in /opt/rocm's copy and the second runtime then finds no device.
    try:
        import torch  # noqa: F401
        rtc = os.path.join(os.path.dirname(torch.__file__), "lib", "libhiprtc.so")
        if os.path.exists(rtc):  # JIT with the hiprtc that matches the HIP runtime torch brought into this process
            os.environ.setdefault("LDPC_HIP_HIPRTC_PATH", rtc)
        rccl = os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")
        if os.path.exists(rccl):  # likewise one RCCL per process (ldpc_hip_open_multi dlopen()s it)
            os.environ.setdefault("LDPC_HIP_RCCL_PATH", rccl)
    except ImportError:
        pass
    path = library_path()
    if not os.path.exists(path):
        raise LdpcHipError(f"{path} not found: run `python -c 'import __graft_entry__ as g; g.build()'` (hipcc, gfx950). "
                           "There is no CPU fallback.")
    lib = C.CDLL(path)
    vp, i32, i64, f64, u64 = C.c_void_p, C.c_int, C.c_longlong, C.c_double, C.c_uint64
    lib.ldpc_hip_abi_version.restype = i32
    lib.ldpc_hip_last_error.restype = C.c_char_p
    lib.ldpc_hip_device_count.restype = i32
    lib.ldpc_hip_open.argtypes = [i32, i32, i32, i32, vp, i32, C.POINTER(vp)]
    lib.ldpc_hip_close.argtypes = [vp]
    lib.ldpc_hip_close.restype = None
    for f in ("ldpc_hip_n", "ldpc_hip_r", "ldpc_hip_edges", "ldpc_hip_hard_words"):
        getattr(lib, f).argtypes = [vp]
    lib.ldpc_hip_kernel_name.argtypes = [vp]
    lib.ldpc_hip_last_launch.argtypes = [vp]
    lib.ldpc_hip_last_launch.restype = C.c_char_p
    lib.ldpc_hip_kernel_name.restype = C.c_char_p
    lib.ldpc_hip_decode_dev.argtypes = [vp, vp, i64, i32, f64, vp, vp, vp, vp]
    lib.ldpc_hip_decode_host.argtypes = [vp, vp, i64, i32, i32, f64, vp, vp, i32]
    lib.ldpc_hip_awgn_llr_dev.argtypes = [vp, f64, i32, i32, u64, i64, i64, vp, vp]
    lib.ldpc_hip_awgn_qam16_llr_dev.argtypes = [vp, f64, f64, u64, i64, i64, vp, vp]
    lib.ldpc_hip_awgn_qam_llr_dev.argtypes = [vp, i32, f64, f64, u64, i64, i64, vp, vp]
    lib.ldpc_hip_qam_demod_dev.argtypes = [i32, f64, f64, vp, i64, vp, i32, i32, vp]
    lib.ldpc_hip_count_errors_dev.argtypes = [vp, vp, vp, i64, vp, vp, vp]
    lib.ldpc_hip_simulate.argtypes = [vp, f64, i32, i32, i32, f64, u64, i64, i64, C.POINTER(C.c_ulonglong), C.POINTER(C.c_ulonglong)]
    lib.ldpc_hip_set_bp_chain.argtypes = [vp, i32, i32]
    lib.ldpc_hip_set_ims_params.argtypes = [vp, f64, i32, i32]
    lib.ldpc_hip_encode_host.argtypes = [i32, i32, i32, vp, vp, vp]
    lib.ldpc_hip_interleaver_build.argtypes = [i32, i32, i32, i32, i32, i32, i32, vp, vp, vp]
    lib.ldpc_hip_permute_dev.argtypes = [vp, vp, i64, i32, vp, i32, vp]
    lib.ldpc_hip_profile_enable.argtypes = [vp, i32]
    lib.ldpc_hip_profile_read.argtypes = [vp, C.POINTER(f64), C.POINTER(i64), i32]
    lib.ldpc_hip_set_interleaver.argtypes = [vp, i32, i32, i32]
    lib.ldpc_hip_set_codewords.argtypes = [vp, vp, i32]
    lib.ldpc_hip_channel_llr_dev.argtypes = [vp, f64, i32, i32, f64, u64, i64, i64, vp, vp]
    lib.ldpc_hip_qam_modulate_dev.argtypes = [i32, vp, i64, vp, i32, vp]
    lib.ldpc_hip_count_errors_cw_dev.argtypes = [vp, vp, vp, i64, i64, vp, vp, vp]
    lib.ldpc_hip_open_multi.argtypes = [i32, i32, i32, i32, vp, vp, i32, C.POINTER(vp)]
    lib.ldpc_hip_close_multi.argtypes = [vp]
    lib.ldpc_hip_close_multi.restype = None
    lib.ldpc_hip_multi_shards.argtypes = [vp]
    lib.ldpc_hip_multi_ctx.argtypes = [vp, i32]
    lib.ldpc_hip_multi_ctx.restype = vp
    lib.ldpc_hip_multi_reduction.argtypes = [vp]
    lib.ldpc_hip_multi_reduction.restype = C.c_char_p
    lib.ldpc_hip_multi_set_interleaver.argtypes = [vp, i32, i32, i32]
    lib.ldpc_hip_multi_set_codewords.argtypes = [vp, vp, i32]
    lib.ldpc_hip_simulate_multi.argtypes = [vp, f64, i32, i32, i32, f64, u64, i64, i64, i64, C.POINTER(C.c_ulonglong), C.POINTER(C.c_ulonglong)]
    lib.ldpc_hip_frames_multi.argtypes = [vp, f64, i32, i32, i32, f64, u64, i64, i64, i64, vp, vp, C.POINTER(C.c_ulonglong), C.POINTER(C.c_ulonglong)]
    lib.ldpc_hip_decode_host_multi.argtypes = [vp, vp, i64, i32, i32, f64, vp, vp, i32]
    lib.ldpc_hip_set_jit_mode.argtypes = [i32]
    lib.ldpc_hip_encode_dev.argtypes = [vp, vp, i64, vp, vp]
    lib.ldpc_hip_set_random_codewords.argtypes = [vp, u64, i32]
    lib.ldpc_hip_multi_set_random_codewords.argtypes = [vp, u64, i32]
    lib.ldpc_hip_mt_jump_host.argtypes = [vp, i32, vp]
    lib.ldpc_hip_mt_set_state.argtypes = [vp, vp, i32]
    lib.ldpc_hip_mt_get_state.argtypes = [vp, vp, C.POINTER(i32)]
    lib.ldpc_hip_mt_normal_dev.argtypes = [vp, i64, vp, vp]
    lib.ldpc_hip_mt_normal_host.argtypes = [vp, i64, vp]
    lib.ldpc_hip_mt_llr_dev.argtypes = [vp, f64, i32, i32, i64, vp, vp]
    lib.ldpc_hip_mt_frames.argtypes = [vp, f64, i32, i32, i32, f64, i64, vp, vp]
    lib.ldpc_hip_mt_frames_slice.argtypes = [vp, f64, i32, i32, i32, f64, i64, i64, i64, vp, vp]
    lib.ldpc_hip_set_jit_mode_thread.argtypes = [i32]
    lib.ldpc_hip_mt_set_frame_index.argtypes = [vp, i64]
    lib.ldpc_hip_mt_get_frame_index.argtypes = [vp]
    lib.ldpc_hip_mt_get_frame_index.restype = i64
    lib.ldpc_hip_multi_stream.argtypes = [vp, i32]
    lib.ldpc_hip_multi_stream.restype = vp
    lib.ldpc_hip_multi_comm_inits.restype = i64
    lib.ldpc_hip_multi_release_comms.restype = None
    lib.ldpc_hip_decode_count_multi.argtypes = [vp, vp, i64, i64, i32, f64, C.POINTER(C.c_ulonglong), C.POINTER(C.c_ulonglong)]
    lib.ldpc_hip_mt_set_state_multi.argtypes = [vp, vp, i32]
    lib.ldpc_hip_mt_get_state_multi.argtypes = [vp, vp, C.POINTER(i32)]
    lib.ldpc_hip_mt_set_frame_index_multi.argtypes = [vp, i64]
    lib.ldpc_hip_mt_advance_multi.argtypes = [vp, f64, i32, i32, i64]
    lib.ldpc_hip_mt_frames_multi.argtypes = [vp, f64, i32, i32, i32, f64, i64, vp, vp]
    lib.ldpc_hip_mt_shard_begin.argtypes = [vp, f64, i32, i32, i64, i32, i32, C.POINTER(C.c_ulonglong)]
    lib.ldpc_hip_mt_shard_emit.argtypes = [vp, vp, C.POINTER(i32), C.POINTER(i32), vp, C.POINTER(i64)]
    lib.ldpc_hip_mt_shard_commit.argtypes = [vp, vp, i64, i32, f64, vp, vp]
    lib.ldpc_hip_mt_shard_abandon.argtypes = [vp]
    lib.ldpc_hip_mt_shard_abandon.restype = None
    lib.ldpc_hip_multi_mt_stats.argtypes = [vp, C.POINTER(i64), C.POINTER(i64)]
    lib.ldpc_hip_multi_mt_stats.restype = None
    if lib.ldpc_hip_abi_version() != 4:
        raise LdpcHipError("libldpc_hip.so ABI version mismatch")
    _lib = lib
    return lib


def _check(lib, rc, what):
    if rc != 0:
        raise LdpcHipError(f"{what}: {lib.ldpc_hip_last_error().decode()} (code {rc})")


def _stream_ptr(stream):
    if stream is None:
        import torch
        return C.c_void_p(torch.cuda.current_stream().cuda_stream)
    return C.c_void_p(getattr(stream, "cuda_stream", stream))


class LdpcHip:
    """One opened code on one GPU == upstream's DEC_STATE (decod_open + hd fill + decod_init)."""

    def __init__(self, decoder_id, H, M, device=0):
        self.lib = load_library()
        H = np.ascontiguousarray(H, dtype=np.int16)
        self.rh, self.nh = H.shape
        self.M, self.decoder_id, self.device = int(M), int(decoder_id), int(device)
        h = C.c_void_p()
        rc = self.lib.ldpc_hip_open(self.decoder_id, self.rh, self.nh, self.M, H.ctypes.data, self.device, C.byref(h))
        _check(self.lib, rc, "ldpc_hip_open")
        self.h = h
        self.N = self.lib.ldpc_hip_n(h)
        self.R = self.lib.ldpc_hip_r(h)
        self.edges = self.lib.ldpc_hip_edges(h)
        self.hard_words = self.lib.ldpc_hip_hard_words(h)
        self._kernel_name = self.lib.ldpc_hip_kernel_name(h).decode()

    @property
    def kernel_name(self):
        """Kernel this context launches; with LDPC_HIP_JIT=async it changes once the background hiprtc instance is ready."""
        if getattr(self, "h", None):
            self._kernel_name = self.lib.ldpc_hip_kernel_name(self.h).decode()
        return self._kernel_name

    def close(self):
        if getattr(self, "h", None):
            self.lib.ldpc_hip_close(self.h)
            self.h = None

    def last_launch(self):
        return self.lib.ldpc_hip_last_launch(self.h).decode()

    def set_ims_params(self, thr=1.4, qbits=6, dbits=8):
        """IMS_DEC: the quantiser arguments of imin_sum_decod_qc_lm (defaults MS_THR, MS_QBITS, MS_DBITS of decoders.h:46-48)."""
        _check(self.lib, self.lib.ldpc_hip_set_ims_params(self.h, float(thr), int(qbits), int(dbits)), "ldpc_hip_set_ims_params")

    def set_bp_chain(self, on=True, reset_carry=False):
        """BP_DEC: chain frames through upstream's uncleared syndrome array (include/ldpc_hip.h)."""
        _check(self.lib, self.lib.ldpc_hip_set_bp_chain(self.h, int(on), int(reset_carry)), "ldpc_hip_set_bp_chain")

    def set_interleaver(self, permutation_type, permutation_block=128, permutation_inter=1):
        """Bit interleaver between mapper / demapper and the code (permutation_type 0..4 of bp_simulation.h:21-23)."""
        _check(self.lib, self.lib.ldpc_hip_set_interleaver(self.h, int(permutation_type), int(permutation_block), int(permutation_inter)),
               "ldpc_hip_set_interleaver")

    def set_codewords(self, codewords):
        """Transmitted codewords uint8 [C, N] (frame f carries codeword f % C); None or empty = upstream's all-zero codeword."""
        if codewords is None or len(codewords) == 0:
            _check(self.lib, self.lib.ldpc_hip_set_codewords(self.h, None, 0), "ldpc_hip_set_codewords")
            return
        cw = np.ascontiguousarray(codewords, dtype=np.uint8).reshape(-1, self.N)
        _check(self.lib, self.lib.ldpc_hip_set_codewords(self.h, cw.ctypes.data, cw.shape[0]), "ldpc_hip_set_codewords")

    def set_random_codewords(self, seed, ncw):
        """ncw random codewords made on the device (Philox information bits + the device encoder); frame f carries codeword f % ncw."""
        _check(self.lib, self.lib.ldpc_hip_set_random_codewords(self.h, int(seed), int(ncw)), "ldpc_hip_set_random_codewords")

    def encode_dev(self, info_bits, stream=None):
        """uint8 CUDA tensor [B, (nh-rh)*M] of 0/1 -> codewords uint8 [B, N] (parity first), on the device."""
        import torch
        info = info_bits.contiguous()
        assert info.dtype == torch.uint8 and info.shape[1] == self.N - self.R
        out = torch.empty((info.shape[0], self.N), dtype=torch.uint8, device=info.device)
        rc = self.lib.ldpc_hip_encode_dev(self.h, C.c_void_p(info.data_ptr()), int(info.shape[0]), C.c_void_p(out.data_ptr()), _stream_ptr(stream))
        _check(self.lib, rc, "ldpc_hip_encode_dev")
        return out

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    # ---- device-resident batch API -------------------------------------------------------------------
    def _dev(self):
        import torch
        return torch.device("cuda", self.device)

    def decode(self, llr, maxiter, alpha=0.8, want_hard=True, want_soft=False, stream=None, out=None):
        """llr: torch float64 CUDA tensor [B,N] (not modified).  Returns (hard int32[B,W] | None, iters int32[B], soft | None).
        hard holds the packed uint32 words reinterpreted as int32 (torch has no uint32 arithmetic)."""
        import torch
        assert llr.is_cuda and llr.dtype == torch.float64 and llr.is_contiguous() and llr.shape[-1] == self.N
        B = llr.shape[0]
        if out is not None:
            hard, iters, soft = out
        else:
            hard = torch.empty((B, self.hard_words), dtype=torch.int32, device=llr.device) if want_hard else None
            iters = torch.empty((B,), dtype=torch.int32, device=llr.device)
            soft = torch.empty((B, self.N), dtype=torch.float64, device=llr.device) if want_soft else None
        rc = self.lib.ldpc_hip_decode_dev(self.h, llr.data_ptr(), B, int(maxiter), float(alpha),
                                          hard.data_ptr() if hard is not None else None, iters.data_ptr(),
                                          soft.data_ptr() if soft is not None else None, _stream_ptr(stream))
        _check(self.lib, rc, "ldpc_hip_decode_dev")
        return hard, iters, soft

    def awgn_llr(self, snr_db, seed, first_frame, B, modulation=0, punctured_blocks=0, out=None, stream=None, T=26.0):
        """Device-side channel: modulation 0 BPSK, 1 QAM4 (upstream formulas), 2 / 3 / 4 the 16- / 64- / 256-QAM chain (as intended upstream)."""
        import torch
        llr = out if out is not None else torch.empty((B, self.N), dtype=torch.float64, device=self._dev())
        rc = self.lib.ldpc_hip_channel_llr_dev(self.h, float(snr_db), int(modulation), int(punctured_blocks), float(T), int(seed),
                                               int(first_frame), int(B), llr.data_ptr(), _stream_ptr(stream))
        _check(self.lib, rc, "ldpc_hip_channel_llr_dev")
        return llr

    channel_llr = awgn_llr   # the whole chain: codeword -> interleaver -> mapper -> AWGN -> demapper -> interleaver -> puncturing

    def count_errors(self, hard, iters, counters=None, want_frame_info=False, stream=None, first_frame=0):
        """counters: int64 CUDA tensor[5] accumulated in place: nse, nde, nue, frames, sum|iters| (bp_simulation.cpp:805-810);
        hard decisions are compared with the codeword each frame carried (first_frame = global index of frame 0)."""
        import torch
        B = iters.shape[0]
        if counters is None:
            counters = torch.zeros(5, dtype=torch.int64, device=iters.device)
        info = torch.empty((B,), dtype=torch.int32, device=iters.device) if want_frame_info else None
        rc = self.lib.ldpc_hip_count_errors_cw_dev(self.h, hard.data_ptr(), iters.data_ptr(), int(first_frame), B,
                                                   info.data_ptr() if info is not None else None, counters.data_ptr(),
                                                   _stream_ptr(stream))
        _check(self.lib, rc, "ldpc_hip_count_errors_cw_dev")
        return counters, info

    def simulate(self, snr_db, maxiter, seed, first_frame, B, modulation=0, punctured_blocks=0, alpha=0.8):
        cnt = (C.c_ulonglong * 4)()
        sit = C.c_ulonglong()
        rc = self.lib.ldpc_hip_simulate(self.h, float(snr_db), int(modulation), int(punctured_blocks), int(maxiter),
                                        float(alpha), int(seed), int(first_frame), int(B), cnt, C.byref(sit))
        _check(self.lib, rc, "ldpc_hip_simulate")
        return {"nse": cnt[0], "nde": cnt[1], "nue": cnt[2], "frames": cnt[3], "sum_abs_iters": sit.value}

    # ---- upstream's own mt19937 / normal_distribution noise stream on the device (exact replay) ----------
    def mt_set_state(self, state, pos):
        """state: 624 uint32 words + index of the next word, as libstdc++ streams a std::mt19937."""
        st = np.ascontiguousarray(state, dtype=np.uint32)
        assert st.shape == (624,)
        _check(self.lib, self.lib.ldpc_hip_mt_set_state(self.h, st.ctypes.data, int(pos)), "ldpc_hip_mt_set_state")

    def mt_get_state(self):
        st = np.empty(624, dtype=np.uint32)
        pos = C.c_int()
        _check(self.lib, self.lib.ldpc_hip_mt_get_state(self.h, st.ctypes.data, C.byref(pos)), "ldpc_hip_mt_get_state")
        return st, pos.value

    def mt_frame_index(self):
        """Frames drawn since mt_set_state (which codeword a frame carries: f % ncw); part of a generator snapshot."""
        return int(self.lib.ldpc_hip_mt_get_frame_index(self.h))

    def mt_set_frame_index(self, frames_taken):
        _check(self.lib, self.lib.ldpc_hip_mt_set_frame_index(self.h, int(frames_taken)), "ldpc_hip_mt_set_frame_index")

    def mt_normal(self, count, out=None, skip=False, stream=None):
        """The next `count` values of upstream's next_random_gaussian() (float64 CUDA tensor); skip=True draws and drops them."""
        import torch
        if not skip and out is None:
            out = torch.empty(int(count), dtype=torch.float64, device=self._dev())
        rc = self.lib.ldpc_hip_mt_normal_dev(self.h, int(count), None if skip else C.c_void_p(out.data_ptr()), _stream_ptr(stream))
        _check(self.lib, rc, "ldpc_hip_mt_normal_dev")
        return out

    def mt_llr(self, snr_db, B, modulation=0, punctured_blocks=0, out=None, skip=False, stream=None):
        """Decoder input of the next B frames of upstream's frame loop ([B,N] float64 CUDA tensor); skip=True only advances."""
        import torch
        if not skip and out is None:
            out = torch.empty((int(B), self.N), dtype=torch.float64, device=self._dev())
        rc = self.lib.ldpc_hip_mt_llr_dev(self.h, float(snr_db), int(modulation), int(punctured_blocks), int(B),
                                          None if skip else C.c_void_p(out.data_ptr()), _stream_ptr(stream))
        _check(self.lib, rc, "ldpc_hip_mt_llr_dev")
        return out

    def mt_frames(self, snr_db, maxiter, B, modulation=0, punctured_blocks=0, alpha=0.8, lo=None, hi=None):
        """noise -> decode -> count for the next B frames: (frame_info, iters) int32 numpy arrays.  With lo / hi the generator still
        advances by all B frames but only frames [lo, hi) are decoded here (one rank's share when every rank runs the generator)."""
        lo = 0 if lo is None else int(lo)
        hi = int(B) if hi is None else int(hi)
        info = np.empty(max(hi - lo, 0), dtype=np.int32)
        its = np.empty(max(hi - lo, 0), dtype=np.int32)
        rc = self.lib.ldpc_hip_mt_frames_slice(self.h, float(snr_db), int(modulation), int(punctured_blocks), int(maxiter), float(alpha), int(B),
                                               lo, hi, info.ctypes.data, its.ctypes.data)
        _check(self.lib, rc, "ldpc_hip_mt_frames_slice")
        return info, its

    # the generator's tape shared out over the ranks of a job, one process per GPU (ldpc_hip_mt_shard_*: the exchanges are the caller's)
    def mt_shard_begin(self, snr_db, frames, rank, n, modulation=0, punctured_blocks=0):
        own = C.c_ulonglong()
        _check(self.lib, self.lib.ldpc_hip_mt_shard_begin(self.h, float(snr_db), int(modulation), int(punctured_blocks), int(frames), int(rank), int(n),
                                                          C.byref(own)), "ldpc_hip_mt_shard_begin")
        return own.value

    def mt_shard_emit(self, counts):
        cnt = np.ascontiguousarray(counts, dtype=np.uint64)
        found, covered, fdone = C.c_int(), C.c_int(), C.c_longlong()
        st = np.zeros(624, dtype=np.uint32)
        _check(self.lib, self.lib.ldpc_hip_mt_shard_emit(self.h, cnt.ctypes.data, C.byref(found), C.byref(covered), st.ctypes.data, C.byref(fdone)),
               "ldpc_hip_mt_shard_emit")
        return bool(found.value), bool(covered.value), st, fdone.value

    def mt_shard_commit(self, state, frames_done, maxiter, rows, alpha=0.8):
        st = np.ascontiguousarray(state, dtype=np.uint32)
        info = np.empty(max(int(rows), 0), dtype=np.int32)
        its = np.empty(max(int(rows), 0), dtype=np.int32)
        _check(self.lib, self.lib.ldpc_hip_mt_shard_commit(self.h, st.ctypes.data, int(frames_done), int(maxiter), float(alpha),
                                                           info.ctypes.data if rows > 0 else None, its.ctypes.data if rows > 0 else None), "ldpc_hip_mt_shard_commit")
        return info, its

    def mt_shard_abandon(self):
        self.lib.ldpc_hip_mt_shard_abandon(self.h)

    # ---- host-pointer API (upstream array layout, PCIe inclusive) --------------------------------------
    def decode_host(self, llr, maxiter, decision=0, alpha=0.8, clobber_sp_input=True):
        """llr: float64 [B,N] or [N].  Returns (decword, iters, llr_after) like the upstream decoder call: decword is
        0.0/1.0 (decision 0) or the a-posteriori values (decision 1); llr_after is what upstream leaves in its input
        array (unchanged for MS/LMS, the likelihood ratios for SP, decoders.cpp:1950,2124)."""
        llr = np.array(llr, dtype=np.float64, order="C", copy=True)
        single = llr.ndim == 1
        if single:
            llr = llr[None, :]
        B = llr.shape[0]
        assert llr.shape[1] == self.N
        dec = np.empty((B, self.N), dtype=np.float64)
        its = np.empty(B, dtype=np.int32)
        rc = self.lib.ldpc_hip_decode_host(self.h, llr.ctypes.data, B, int(maxiter), int(decision), float(alpha),
                                           dec.ctypes.data, its.ctypes.data, 1 if clobber_sp_input else 0)
        _check(self.lib, rc, "ldpc_hip_decode_host")
        if single:
            return dec[0], int(its[0]), llr[0]
        return dec, its, llr

    # ---- kernel timing (HIP events on the launch stream) ------------------------------------------------
    def profile(self, enable=True):
        _check(self.lib, self.lib.ldpc_hip_profile_enable(self.h, 1 if enable else 0), "ldpc_hip_profile_enable")

    def profile_read(self, reset=True):
        ms, n = C.c_double(), C.c_longlong()
        _check(self.lib, self.lib.ldpc_hip_profile_read(self.h, C.byref(ms), C.byref(n), 1 if reset else 0), "ldpc_hip_profile_read")
        return ms.value, n.value


class LdpcHipMulti:
    """One opened code on several GPUs of a node behind the C-ABI (ldpc_hip_open_multi): one context + stream + host thread per
    shard, frames sharded by global frame index, the five counters all-reduced over RCCL (or summed on the host when shards
    share a device)."""

    def __init__(self, decoder_id, H, M, devices):
        self.lib = load_library()
        H = np.ascontiguousarray(H, dtype=np.int16)
        self.rh, self.nh = H.shape
        devs = np.ascontiguousarray(devices, dtype=np.int32)
        h = C.c_void_p()
        rc = self.lib.ldpc_hip_open_multi(int(decoder_id), self.rh, self.nh, int(M), H.ctypes.data, devs.ctypes.data, len(devs), C.byref(h))
        _check(self.lib, rc, "ldpc_hip_open_multi")
        self.h = h
        self.N, self.R = self.nh * int(M), self.rh * int(M)
        self.shards = self.lib.ldpc_hip_multi_shards(h)
        self.reduction = self.lib.ldpc_hip_multi_reduction(h).decode()

    def close(self):
        if getattr(self, "h", None):
            self.lib.ldpc_hip_close_multi(self.h)
            self.h = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_interleaver(self, permutation_type, permutation_block=128, permutation_inter=1):
        _check(self.lib, self.lib.ldpc_hip_multi_set_interleaver(self.h, int(permutation_type), int(permutation_block), int(permutation_inter)),
               "ldpc_hip_multi_set_interleaver")

    def set_codewords(self, codewords):
        if codewords is None or len(codewords) == 0:
            _check(self.lib, self.lib.ldpc_hip_multi_set_codewords(self.h, None, 0), "ldpc_hip_multi_set_codewords")
            return
        cw = np.ascontiguousarray(codewords, dtype=np.uint8).reshape(-1, self.N)
        _check(self.lib, self.lib.ldpc_hip_multi_set_codewords(self.h, cw.ctypes.data, cw.shape[0]), "ldpc_hip_multi_set_codewords")

    def simulate(self, snr_db, maxiter, seed, first_frame, B, batch, modulation=0, punctured_blocks=0, alpha=0.8, records=False):
        cnt = (C.c_ulonglong * 4)()
        sit = C.c_ulonglong()
        if records:
            info = np.empty(B, dtype=np.int32)
            its = np.empty(B, dtype=np.int32)
            rc = self.lib.ldpc_hip_frames_multi(self.h, float(snr_db), int(modulation), int(punctured_blocks), int(maxiter), float(alpha),
                                                int(seed), int(first_frame), int(B), int(batch), info.ctypes.data, its.ctypes.data, cnt, C.byref(sit))
            _check(self.lib, rc, "ldpc_hip_frames_multi")
        else:
            info = its = None
            rc = self.lib.ldpc_hip_simulate_multi(self.h, float(snr_db), int(modulation), int(punctured_blocks), int(maxiter), float(alpha),
                                                  int(seed), int(first_frame), int(B), int(batch), cnt, C.byref(sit))
            _check(self.lib, rc, "ldpc_hip_simulate_multi")
        out = {"nse": cnt[0], "nde": cnt[1], "nue": cnt[2], "frames": cnt[3], "sum_abs_iters": sit.value}
        if records:
            out["frame_info"], out["iters"] = info, its
        return out

    def ctx(self, shard):
        """Raw handle of shard i's context (per-shard settings and the single-device entry points)."""
        return C.c_void_p(self.lib.ldpc_hip_multi_ctx(self.h, int(shard)))

    def stream(self, shard):
        return C.c_void_p(self.lib.ldpc_hip_multi_stream(self.h, int(shard)))

    def channel_llr(self, shard, out, snr_db, seed, first_frame, modulation=0, punctured_blocks=0, T=26.0):
        """Fills `out` (float64 CUDA tensor [B, N] on the shard's device) with the decoder input of frames [first_frame, first_frame + B),
        on the shard's stream."""
        rc = self.lib.ldpc_hip_channel_llr_dev(self.ctx(shard), float(snr_db), int(modulation), int(punctured_blocks), float(T), int(seed),
                                               int(first_frame), int(out.shape[0]), C.c_void_p(out.data_ptr()), self.stream(shard))
        _check(self.lib, rc, "ldpc_hip_channel_llr_dev")
        return out

    def decode_count(self, batches, maxiter, first_frame=0, alpha=0.8):
        """The hot path on batches resident on the GPUs: batches[i] = float64 CUDA tensor [B, N] on shard i's device.  Returns the
        all-reduced counters."""
        assert len(batches) == self.shards
        B = int(batches[0].shape[0])
        ptrs = (C.c_void_p * self.shards)(*[C.c_void_p(b.data_ptr()) for b in batches])
        cnt = (C.c_ulonglong * 4)()
        sit = C.c_ulonglong()
        rc = self.lib.ldpc_hip_decode_count_multi(self.h, ptrs, B, int(first_frame), int(maxiter), float(alpha), cnt, C.byref(sit))
        _check(self.lib, rc, "ldpc_hip_decode_count_multi")
        return {"nse": cnt[0], "nde": cnt[1], "nue": cnt[2], "frames": cnt[3], "sum_abs_iters": sit.value}

    def mt_set_state(self, state, pos):
        st = np.ascontiguousarray(state, dtype=np.uint32)
        _check(self.lib, self.lib.ldpc_hip_mt_set_state_multi(self.h, st.ctypes.data, int(pos)), "ldpc_hip_mt_set_state_multi")

    def mt_get_state(self):
        st = np.empty(624, dtype=np.uint32)
        pos = C.c_int()
        _check(self.lib, self.lib.ldpc_hip_mt_get_state_multi(self.h, st.ctypes.data, C.byref(pos)), "ldpc_hip_mt_get_state_multi")
        return st, pos.value

    def mt_frames(self, snr_db, maxiter, B, modulation=0, punctured_blocks=0, alpha=0.8):
        info = np.empty(int(B), dtype=np.int32)
        its = np.empty(int(B), dtype=np.int32)
        rc = self.lib.ldpc_hip_mt_frames_multi(self.h, float(snr_db), int(modulation), int(punctured_blocks), int(maxiter), float(alpha), int(B),
                                               info.ctypes.data, its.ctypes.data)
        _check(self.lib, rc, "ldpc_hip_mt_frames_multi")
        return info, its

    def mt_stats(self):
        """(generation rounds shared out over the shards, rounds redone with the whole tape on every shard)"""
        a, b = C.c_longlong(), C.c_longlong()
        self.lib.ldpc_hip_multi_mt_stats(self.h, C.byref(a), C.byref(b))
        return a.value, b.value

    def mt_set_frame_index(self, frames_taken):
        _check(self.lib, self.lib.ldpc_hip_mt_set_frame_index_multi(self.h, int(frames_taken)), "ldpc_hip_mt_set_frame_index_multi")

    def mt_advance(self, snr_db, B, modulation=0, punctured_blocks=0):
        _check(self.lib, self.lib.ldpc_hip_mt_advance_multi(self.h, float(snr_db), int(modulation), int(punctured_blocks), int(B)), "ldpc_hip_mt_advance_multi")

    def decode_host(self, llr, maxiter, decision=0, alpha=0.8, clobber_sp_input=True):
        llr = np.array(llr, dtype=np.float64, order="C", copy=True)
        B = llr.shape[0]
        dec = np.empty((B, self.N), dtype=np.float64)
        its = np.empty(B, dtype=np.int32)
        rc = self.lib.ldpc_hip_decode_host_multi(self.h, llr.ctypes.data, B, int(maxiter), int(decision), float(alpha), dec.ctypes.data,
                                                 its.ctypes.data, 1 if clobber_sp_input else 0)
        _check(self.lib, rc, "ldpc_hip_decode_host_multi")
        return dec, its, llr


def mt_jump_host(state, log2_words):
    """mt19937 state 2**log2_words words further on (host, no GPU): the library's GF(2) polynomial jump."""
    lib = load_library()
    st = np.ascontiguousarray(state, dtype=np.uint32)
    out = np.empty(624, dtype=np.uint32)
    _check(lib, lib.ldpc_hip_mt_jump_host(st.ctypes.data, int(log2_words), out.ctypes.data), "ldpc_hip_mt_jump_host")
    return out


def qam_modulate(bits, Q, device=0, stream=None):
    """Function-level mapper (QAM_modulator.cpp QAM_modulator): uint8 CUDA tensor bits[ns, log2 Q] -> float64 [ns, 2] (I, Q)."""
    import torch
    lib = load_library()
    ns = bits.shape[0]
    out = torch.empty((ns, 2), dtype=torch.float64, device=bits.device)
    rc = lib.ldpc_hip_qam_modulate_dev(int(Q), bits.data_ptr(), ns, out.data_ptr(), device, _stream_ptr(stream))
    _check(lib, rc, "ldpc_hip_qam_modulate_dev")
    return out


def encode(H, M, info_bits):
    """Systematic QC-LDPC encoding on the host (upstream's qc_encode / random_codeword with given information bits):
    info_bits uint8[(nh-rh)*M] -> codeword uint8[nh*M] (parity part first).  Needs no GPU."""
    lib = load_library()
    H = np.ascontiguousarray(H, dtype=np.int16)
    rh, nh = H.shape
    info = np.ascontiguousarray(info_bits, dtype=np.uint8)
    assert info.size == (nh - rh) * M
    cw = np.empty(nh * M, dtype=np.uint8)
    _check(lib, lib.ldpc_hip_encode_host(rh, nh, int(M), H.ctypes.data, info.ctypes.data, cw.ctypes.data), "ldpc_hip_encode_host")
    return cw


def build_interleaver(H, M, mode, halfmlog=1, block_size=128, step_size=1):
    """Gather maps (direct, inverse) of upstream's interleaver `mode` (permutation_type) for base matrix H: out[i] = in[map[i]].
    Host-side, needs no GPU."""
    lib = load_library()
    H = np.ascontiguousarray(H, dtype=np.int16)
    b, c = H.shape
    direct = np.empty(c * M, dtype=np.int32)
    inverse = np.empty(c * M, dtype=np.int32)
    rc = lib.ldpc_hip_interleaver_build(b, c, int(M), int(halfmlog), int(mode), int(block_size), int(step_size), H.ctypes.data,
                                        direct.ctypes.data, inverse.ctypes.data)
    _check(lib, rc, "ldpc_hip_interleaver_build")
    return direct, inverse


def permute(x, index_map, device=0, stream=None):
    """y[f, i] = x[f, index_map[i]] on the GPU (x float64 CUDA [B, N], index_map int32 CUDA [N])."""
    import torch
    lib = load_library()
    y = torch.empty_like(x)
    rc = lib.ldpc_hip_permute_dev(x.data_ptr(), y.data_ptr(), x.shape[0], x.shape[1], index_map.data_ptr(), device, _stream_ptr(stream))
    _check(lib, rc, "ldpc_hip_permute_dev")
    return y


def qam_demod(x, Q, T, sigma, out_type=0, device=0, stream=None):
    """Function-level soft demapper (QAM_demodulator.cpp Demodulate) on a CUDA float64 tensor x[ns,2] -> [ns, log2 Q]."""
    import torch
    lib = load_library()
    m = {4: 2, 16: 4, 64: 6, 256: 8}[Q]
    ns = x.shape[0]
    out = torch.empty((ns, m), dtype=torch.float64, device=x.device)
    rc = lib.ldpc_hip_qam_demod_dev(Q, float(T), float(sigma), x.data_ptr(), ns, out.data_ptr(), out_type, device, _stream_ptr(stream))
    _check(lib, rc, "ldpc_hip_qam_demod_dev")
    return out
