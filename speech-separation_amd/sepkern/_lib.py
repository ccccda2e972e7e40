"""ctypes binding of libsepkern.so (include/sepkern.h).

The product path has NO fallback: if the HIP library is missing or a call fails, this raises.
"""
import ctypes as C
import os

# torch bundles its own HIP runtime (libamdhip64): it must be loaded BEFORE libsepkern.so so that
# the dynamic linker binds our library to that same runtime instance (one runtime per process;
# device pointers and streams are shared with torch).
import torch  # noqa: F401,E402

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SEPKERN_LIB") or os.path.join(_HERE, "libsepkern.so")   # SEPKERN_LIB: diagnostic builds

# sk_build_flags() bits (include/sepkern.h): what a library built by `make variant` / `make gemm_variant` reports
BUILD_FLAG_NAMES = {0x1: "TIMING_ONLY (wrong results by construction)", 0x2: "ARITH (another arithmetic than documented)",
                    0x4: "TUNING (same results, other tuning constants)", 0x8: "STAMPS (clock stamps in the recurrence kernels)"}


def build_flag_names(mask):
    return [n for b, n in sorted(BUILD_FLAG_NAMES.items()) if mask & b] + (["unknown 0x%x" % (mask & ~0xf)] if mask & ~0xf else [])

SK_VERSION = 131

_p, _i, _i64, _f, _sz = C.c_void_p, C.c_int, C.c_int64, C.c_float, C.c_size_t

# name -> (restype, argtypes): one entry per symbol declared in include/sepkern.h
PROTOTYPES = {
    "sk_version": (_i, []),
    "sk_last_error": (C.c_char_p, []),
    "sk_build_flags": (C.c_uint, []),
    "sk_device_info": (_i, [C.POINTER(_i), C.POINTER(_i)]),
    "sk_stft": (_i, [_p, _i, _p, _p, _i, _i, _i, _i, _p, _p, _p, _p, _i, _i, _p]),
    "sk_mask_istft": (_i, [_p, _p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _p, _p, _p, _i, _p]),
    "sk_gemm_f32": (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i64, _i64, _i64, _i64, _p]),
    "sk_gemm_workspace_bytes": (_sz, [_i, _i, _i, _i]),
    "sk_gemm_streamk_workspace_bytes": (_sz, []),
    "sk_gemm_last_kernel": (_i, []),
    "sk_gemm_workspace_init": (_i, [_p, _p]),
    "sk_gemm_f32_splitk": (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i64, _i64, _i64, _i64, _i, _p, _i, _p]),
    "sk_gemm_bf16_splitk": (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i64, _i64, _i64, _i64, _i, _p, _p]),
    "sk_gemm_bf16_nt": (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i64, _i64, _i64, _i64, _i, _p, _p]),
    "sk_gemm_bf16_mm": (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i64, _i64, _i64, _i64, _i, _p, _p]),
    "sk_cast_bf16_rows": (_i, [_p, _i, _i, _i, _p, _i, _i, _p]),
    "sk_split_rows": (_i, [_p, _i, _i, _i, _p, _i, _i, _i64, _p]),
    "sk_gemm_pl3_tn": (_i, [_p, _p, _p, _i, _i, _i, _i, _i, _i, _i64, _i64, _i, _i, _i64, _i64, _i64, _i, _p, _p]),
    "sk_pack_rows": (_i, [_p, _p, _p, _i, _i, _i, _p, _i, _p]),
    "sk_unpack_rows": (_i, [_p, _i, _p, _p, _i, _i, _i, _p, _p, _p]),
    "sk_hprev_rows": (_i, [_p, _i, _p, _p, _i, _i, _i, _p, _i, _i, _p]),
    "sk_lstm_workspace_bytes": (_sz, [_i, _i, _i]),
    "sk_lstm_fwd": (_i, [_p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _p]),
    "sk_lstm_bwd": (_i, [_p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _i, _i64, _p, _i, _i, _i, _i, _p]),
    "sk_lstm_status": (_i, [_p, _p]),
    "sk_gate_rows": (_i, [_p, _p, _i, _i, _i, _i, _i, _i, _i, _p]),
    "sk_bn_workspace_bytes": (_sz, [_i, _i]),
    "sk_bn_stats": (_i, [_p, _i, _i, _i64, _p, _p, _p, _p]),
    "sk_bn_update_running": (_i, [_p, _p, _p, _p, _i64, _i, _f, _p, _p]),
    "sk_bn_apply": (_i, [_p, _p, _p, _p, _p, _p, _i, _i, _f, _p]),
    "sk_bn_bwd": (_i, [_p, _p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _f, _p]),
    "sk_bn_fold": (_i, [_p, _p, _p, _p, _p, _p, _f, _i, _i, _p, _i, _p, _p, _p, _p]),
    "sk_bn_unfold_grad": (_i, [_p, _i, _p, _p, _p, _p, _i, _i, _i, _p]),
    "sk_bn_bwd_sums": (_i, [_p, _p, _p, _p, _p, _p, _p, _i, _i, _f, _p]),
    "sk_bn_bwd_apply": (_i, [_p, _p, _p, _p, _p, _p, _p, _p, _i, _i, C.c_double, _f, _p]),
    "sk_colsum": (_i, [_p, _i, _i, _i, _p, _i, _p, _p]),
    "sk_sigmoid_bwd": (_i, [_p, _p, _p, _i64, _p]),
    "sk_pad_rows": (_i, [_p, _i64, _i, _i, _p, _i, _i64, _p]),
    "sk_pit_workspace_bytes": (_sz, [_i, _i, _i]),
    "sk_pit_mse_fwd": (_i, [_p, _p, C.POINTER(_p), _p, _p, _i, _i, _i, _i, _p, _p, _p, _p, _p, _p, _p]),
    "sk_pit_mse_bwd": (_i, [_p, _p, C.POINTER(_p), _p, _p, _p, _p, _i64, _i, _i, _i, _i, _p, _p]),
    "sk_rsh_workspace_bytes": (_sz, [_i, _i, _i]),
    "sk_rsh_loss_fwd": (_i, [_p, _p, _i, C.POINTER(_p), _p, _i, _i, _i, _i, _p, _p, _p, _p, _p, _p]),
    "sk_rsh_loss_bwd": (_i, [_p, _p, _i, C.POINTER(_p), _p, _p, _i, _i, _i, _i, _p, _p]),
    "sk_att_update": (_i, [_p, _p, _p, _i64, _i, _i, _p]),
    "sk_att_update_bwd": (_i, [_p, _p, _p, _p, _i64, _i, _i, _p]),
    "sk_optim_workspace_bytes": (_sz, [_i64]),
    "sk_grad_norm": (_i, [_p, _i64, _f, _p, _p, _p, _p]),
    "sk_clip_adam": (_i, [_p, _p, _p, _p, _i64, _p, _f, _f, _f, _f, _i, _p]),
}

_lib = None


class SepkernError(RuntimeError):
    pass


def load():
    """Load libsepkern.so; raises SepkernError when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise SepkernError(
            "libsepkern.so not found at %s -- build it with `python __graft_entry__.py` "
            "(or make -C speech-separation_amd/csrc); there is no CPU fallback" % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in PROTOTYPES.items():
        fn = getattr(lib, name)          # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    v = lib.sk_version()
    if v != SK_VERSION:
        raise SepkernError("libsepkern.so version %d does not match binding %d; rebuild" % (v, SK_VERSION))
    flags = lib.sk_build_flags()
    if flags and os.environ.get("SEPKERN_ALLOW_DIAGNOSTIC_LIB") != "1":
        # a measurement build (csrc/Makefile: variant / gemm_variant / stamps) lying where the product library is looked for, or
        # named by SEPKERN_LIB: its numbers or its numerics are not the product's -- nothing may run on it by accident
        raise SepkernError("%s is a DIAGNOSTIC build (sk_build_flags() = 0x%x: %s); refusing to load it -- set "
                           "SEPKERN_ALLOW_DIAGNOSTIC_LIB=1 for a measurement script, or rebuild with plain `make -C "
                           "speech-separation_amd/csrc`" % (LIB_PATH, flags, "; ".join(build_flag_names(flags))))
    _lib = lib
    return lib


def library_info():
    """{"path", "build_flags", "build_flag_names", "version"} of the loaded library (bench.py prints it on its line)."""
    lib = load()
    flags = int(lib.sk_build_flags())
    return {"path": os.path.relpath(LIB_PATH, os.path.dirname(os.path.dirname(_HERE))), "version": int(lib.sk_version()),
            "build_flags": flags, "build_flag_names": build_flag_names(flags)}


def check(rc, what=""):
    if rc != 0:
        msg = load().sk_last_error().decode("utf-8", "replace")
        raise SepkernError("%s failed (code %d): %s" % (what or "sepkern call", rc, msg))


def call(name, *args):
    """Invoke an int-returning entry point and raise on a non-zero code."""
    rc = getattr(load(), name)(*args)
    check(rc, name)
