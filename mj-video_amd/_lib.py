"""ctypes binding of ``libmjv_hip.so`` (C ABI declared in include/mjv.h).

The product path has NO fallback: if the shared library is missing or an entry point fails, a
``RuntimeError`` is raised (``MjvLibraryError``).  ``build_library()`` compiles it in-tree with hipcc
for gfx950 (cross-compiles without a GPU).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
import threading

_PKG_DIR = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_PKG_DIR, "libmjv_hip.so")
CSRC_DIR = os.path.join(_PKG_DIR, "csrc")

EPI_BIAS, EPI_BIAS_GELU, EPI_BIAS_RELU, EPI_SCALE_RES, EPI_SILU_MUL, EPI_ROPE_QKV = range(6)


class MjvLibraryError(RuntimeError):
    pass


ABI_VERSION = 7  # MJV_ABI_VERSION of include/mjv.h
FMT_BF16, FMT_MXFP8 = 0, 1   # enum mjv_format


class GemmDesc(C.Structure):
    _fields_ = [("A", C.c_void_p), ("lda", C.c_int64), ("W", C.c_void_p), ("ldw", C.c_int64),
                ("C", C.c_void_p), ("ldc", C.c_int64), ("M", C.c_int32), ("N", C.c_int32), ("K", C.c_int32),
                ("epilogue", C.c_int32), ("bias", C.c_void_p), ("scale", C.c_void_p), ("res", C.c_void_p),
                ("ldr", C.c_int64), ("res_mod", C.c_int32), ("res_off", C.c_int32), ("out_group", C.c_int32),
                ("out_pad", C.c_int32), ("out_rows", C.c_void_p), ("workspace", C.c_void_p),
                ("workspace_bytes", C.c_int64), ("tile", C.c_int32), ("rope_cos", C.c_void_p), ("rope_sin", C.c_void_p),
                ("rope_pos", C.c_void_p), ("rope_q", C.c_void_p), ("rope_k", C.c_void_p), ("rope_ldq", C.c_int64),
                ("rope_ldk", C.c_int64), ("rope_group", C.c_int32),
                # ABI 5: operand / output formats (all zero = bf16)
                ("a_format", C.c_int32), ("w_format", C.c_int32), ("c_format", C.c_int32),
                ("a_scales", C.c_void_p), ("w_scales", C.c_void_p), ("c_scales", C.c_void_p),
                # ABI 5: a norm folded into the GEMM (fp32 vectors)
                ("row_scale", C.c_void_p), ("row_shift", C.c_void_p), ("col_shift", C.c_void_p), ("bias_f32", C.c_void_p)]


class AttnDesc(C.Structure):
    _fields_ = [("Q", C.c_void_p), ("K", C.c_void_p), ("V", C.c_void_p), ("O", C.c_void_p),
                ("ldq", C.c_int64), ("ldk", C.c_int64), ("ldv", C.c_int64), ("ldo", C.c_int64),
                ("q_head_stride", C.c_int32), ("k_head_stride", C.c_int32), ("v_head_stride", C.c_int32),
                ("o_head_stride", C.c_int32), ("cu_seqlens", C.c_void_p), ("n_seqs", C.c_int32),
                ("max_seqlen", C.c_int32), ("n_heads", C.c_int32), ("kv_group", C.c_int32),
                ("head_dim", C.c_int32), ("causal", C.c_int32), ("scale", C.c_float),
                ("score_round_mode", C.c_int32), ("kernel", C.c_int32),
                # ABI 6 (zero = ABI 5): suffix queries, shared key / value prefix
                ("cu_seqlens_q", C.c_void_p), ("max_seqlen_q", C.c_int32), ("prefix_len", C.c_int32),
                ("prefix_k", C.c_void_p), ("prefix_v", C.c_void_p)]


class HeadsDesc(C.Structure):
    _fields_ = [("hr", C.c_void_p), ("hg", C.c_void_p), ("ldh", C.c_int64), ("hidden", C.c_int32),
                ("ga", C.c_void_p), ("gc", C.c_void_p), ("ldg", C.c_int64), ("gate_hidden", C.c_int32),
                ("w_reg", C.c_void_p), ("w_transform", C.c_void_p), ("wa", C.c_void_p), ("ba", C.c_void_p),
                ("wc", C.c_void_p), ("bc", C.c_void_p), ("ls_a", C.c_void_p), ("ls_c", C.c_void_p),
                ("temperature", C.c_float), ("batch", C.c_int32), ("n_obj", C.c_int32), ("n_asp", C.c_int32),
                ("group_offsets", C.c_void_p), ("group_index", C.c_void_p), ("rewards", C.c_void_p),
                ("criteria_gating", C.c_void_p), ("aspect_gating", C.c_void_p), ("aspect_weights", C.c_void_p),
                ("weighted_last", C.c_void_p), ("aspect_scores", C.c_void_p), ("score", C.c_void_p),
                ("packed34", C.c_void_p)]


# every symbol include/mjv.h declares: name -> (restype, argtypes)
_VP, _I64, _I32, _F = C.c_void_p, C.c_int64, C.c_int32, C.c_float
SYMBOLS = {
    "mjv_abi_version": (C.c_int, []),
    "mjv_last_error": (C.c_char_p, []),
    "mjv_arch": (C.c_char_p, []),
    "mjv_gemm_bf16": (C.c_int, [C.POINTER(GemmDesc), _VP]),
    "mjv_gemm_workspace_bytes": (C.c_int64, []),
    "mjv_attention_bf16": (C.c_int, [C.POINTER(AttnDesc), _VP]),
    "mjv_layernorm_bf16": (C.c_int, [_VP, _I64, _VP, _I64, _VP, _VP, _I32, _I32, _F, _I32, _VP]),
    "mjv_rmsnorm_bf16": (C.c_int, [_VP, _I64, _VP, _I64, _VP, _VP, _I32, _I32, _F, _VP]),
    "mjv_mxfp8_scale_bytes": (C.c_int64, [_I64, _I64]),
    "mjv_quantize_mxfp8": (C.c_int, [_VP, _I64, _VP, _I64, _VP, _I32, _I32, _VP]),
    "mjv_layernorm_mxfp8": (C.c_int, [_VP, _I64, _VP, _I64, _VP, _VP, _VP, _I32, _I32, _F, _VP]),
    "mjv_rmsnorm_mxfp8": (C.c_int, [_VP, _I64, _VP, _I64, _VP, _VP, _I32, _I32, _F, _VP]),
    "mjv_row_stats_bf16": (C.c_int, [_VP, _I64, _VP, _VP, _I32, _I32, _F, _VP]),
    "mjv_rope_split_bf16": (C.c_int, [_VP, _I64, _VP, _I64, _VP, _I64, _VP, _VP, _VP, _I32, _I32, _I32, _VP]),
    "mjv_rope_heads_bf16": (C.c_int, [_VP, _I64, _I32, _I32, _I32, _VP, _VP, _VP, _I32, _VP]),
    "mjv_patchify_bf16": (C.c_int, [_VP, _VP, _I64, _I32, _I32, _I32, _VP]),
    "mjv_cls_rows_bf16": (C.c_int, [_VP, _I64, _VP, _VP, _I32, _I32, _I32, _VP]),
    "mjv_embed_gather_bf16": (C.c_int, [_VP, _VP, _I64, _VP, _I64, _I32, _I32, _I32, _I32, _VP]),
    "mjv_reward_heads_bf16": (C.c_int, [C.POINTER(HeadsDesc), _VP]),
    "mjv_resize_normalize_u8": (C.c_int, [_VP, _I32, _I32, _I32, _I32, _I32, _VP, _VP, _I32, _VP, _VP, _I32, _VP, _VP, _I32, _I32,
                                           _I32, C.POINTER(C.c_float), C.POINTER(C.c_float), _VP]),
    "mjv_prof_enable": (C.c_int, [_I32]),
    "mjv_prof_filter": (C.c_int, [C.c_char_p]),
    "mjv_prof_reset": (C.c_int, []),
    "mjv_prof_collect": (C.c_int, []),
    "mjv_prof_count": (C.c_int, []),
    "mjv_prof_get": (C.c_int, [_I32, C.POINTER(C.c_char_p), C.POINTER(C.c_int64), C.POINTER(C.c_double),
                               C.POINTER(C.c_double), C.POINTER(C.c_double)]),
}

# include/mjv_bench.h: exported by the BENCH build only (libmjv_hip_bench.so; tools/ load it by pointing MJV_LIBRARY at it)
BENCH_SYMBOLS = {
    "mjv_bench_gemm_set": (C.c_int, [_I32]),
    "mjv_bench_gemm_stamp_buffer": (C.c_int, [_VP]),
    "mjv_bench_attention_set": (C.c_int, [_I32]),
    "mjv_bench_rmsnorm_prestat": (C.c_int, [_VP, _I64, _VP, _I64, _VP, _VP, _I32, _I32, C.c_float, _VP]),
    "mjv_bench_epilogue_pass": (C.c_int, [_VP, _I64, _VP, _I64, _VP, _VP, _VP, _I64, _I32, _I32, _I32, _VP]),
}
BENCH_LIB_PATH = os.path.join(_PKG_DIR, "libmjv_hip_bench.so")

_lock = threading.Lock()
_lib = None
_lib_path = None   # realpath of the file load_library() actually opened (assert_product_library compares THIS, not the env)


def build_library(force: bool = False, verbose: bool = False) -> str:
    """hipcc --offload-arch=gfx950 build of csrc/*.hip into mj-video_amd/libmjv_hip.so (via csrc/Makefile)."""
    if force and os.path.exists(LIB_PATH):
        os.remove(LIB_PATH)
    r = subprocess.run(["make", "-C", CSRC_DIR, "-j", str(min(8, os.cpu_count() or 1))],
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if verbose or r.returncode != 0:
        print(r.stdout)
    if r.returncode != 0 or not os.path.isfile(LIB_PATH):
        raise MjvLibraryError(f"building {LIB_PATH} failed (exit {r.returncode})")
    return LIB_PATH


def load_library():
    """Loads the library once and type-checks every exported symbol.  Never falls back to anything."""
    global _lib, _lib_path
    with _lock:
        if _lib is not None:
            return _lib
        path = os.environ.get("MJV_LIBRARY") or LIB_PATH   # (tools: the bench / stamp builds of the same sources)
        if path != LIB_PATH:
            import warnings
            warnings.warn(f"MJV_LIBRARY={path}: NOT the product library ({LIB_PATH}) - the bench / stamp builds carry process-wide "
                          "measurement switches and kernel variants that are wrong by construction; scoring entry points refuse "
                          "them (assert_product_library)", RuntimeWarning, stacklevel=2)
        if not os.path.isfile(path):
            raise MjvLibraryError(
                f"{path} not found: the HIP extension is required (python -c 'import __graft_entry__ as g; "
                f"g.build()' or make -C {CSRC_DIR}); there is no CPU fallback")
        try:
            lib = C.CDLL(path)
        except OSError as e:
            raise MjvLibraryError(f"cannot load {path}: {e}") from e
        for name, (res, args) in SYMBOLS.items():
            try:
                fn = getattr(lib, name)
            except AttributeError as e:
                raise MjvLibraryError(f"{path} does not export {name}") from e
            fn.restype, fn.argtypes = res, args
        for name, (res, args) in BENCH_SYMBOLS.items():   # present in the bench build only
            fn = getattr(lib, name, None)
            if fn is not None:
                fn.restype, fn.argtypes = res, args
        if lib.mjv_abi_version() != ABI_VERSION:
            raise MjvLibraryError(f"ABI version mismatch: library {lib.mjv_abi_version()} != binding {ABI_VERSION}")
        _lib, _lib_path = lib, os.path.realpath(path)
        return lib


def is_bench_build() -> bool:
    """True when the loaded library exports the measurement switches of include/mjv_bench.h (libmjv_hip_bench.so)"""
    return getattr(getattr(load_library(), "mjv_bench_gemm_set", None), "argtypes", None) is not None


def assert_product_library() -> None:
    """Scoring entry points call this (scripts/eval/* mains, ``harness.score_pair_batch`` / ``score_collated_batch`` /
    ``evaluate_*``): an inherited MJV_LIBRARY must not silently swap a diagnostics build under an evaluation.
    Refused: the bench build (process-wide measurement switches, variants that skip work) AND any library loaded from a path
    other than the in-tree product library - the stamps build exports no bench symbol and would pass a symbol test.
    Direct ``model.forward`` calls (tools/, tests A/B-ing builds) are not policed: they choose their library on purpose."""
    load_library()
    path = _lib_path   # the file that was opened - not what MJV_LIBRARY says now (it may have been changed after the load)
    if is_bench_build() or path != os.path.realpath(LIB_PATH):
        raise MjvLibraryError(f"a diagnostics build of the library is loaded ({path}; MJV_LIBRARY): bench / stamps builds carry "
                              "measurement switches or instrumentation and must not score; unset MJV_LIBRARY")


def check(rc: int, what: str) -> None:
    if rc != 0:
        msg = load_library().mjv_last_error().decode(errors="replace")
        raise MjvLibraryError(f"{what} failed ({rc}): {msg}")
