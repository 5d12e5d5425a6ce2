"""ldpc-lib_amd: MI355X-native batched QC-LDPC belief-propagation decoding behind the call surface of
eovs/ldpc-lib's `decoders.h` / `bp_simulation.h`.

The product is the C-ABI shared library `libldpc_hip.so` (include/ldpc_hip.h) built from csrc/*.hip for gfx950,
plus the C++ source-compatible layer (include/ldpc/*.h, csrc/compat/*.cpp).  This Python package is only the
thin ctypes binding that tests/ and bench.py use (and a reference for other FFI users); PyTorch is used by the
callers for device memory, streams and torch.distributed -- never for the arithmetic.

There is deliberately no CPU fallback: importing works anywhere, but every compute entry point raises
`LdpcHipError` when the HIP library or a GPU is missing.
"""
from .binding import (DEC_ASP, DEC_BP, DEC_IMS, DEC_LMS, DEC_MS, DEC_SP, DEC_TASP, LdpcHip, LdpcHipError, LdpcHipMulti, build_library, library_path,  # noqa: F401
                      load_library)
from .host import GpuFrameSource, MtFrameSource, bp_simulation, mt19937_state, relift_base_matrix, replay_stopping_rule  # noqa: F401

__all__ = ["LdpcHip", "LdpcHipError", "DEC_BP", "DEC_SP", "DEC_ASP", "DEC_MS", "DEC_IMS", "DEC_TASP", "DEC_LMS", "build_library", "library_path",
           "load_library", "bp_simulation", "relift_base_matrix"]
