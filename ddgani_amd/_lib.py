"""ctypes binding of libddgani_hip.so (declared in include/ddgani.h).

There is no CPU fallback: if the library is missing or a call fails, an exception is raised.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("DDGANI_LIB") or os.path.join(_HERE, "libddgani_hip.so")   # DDGANI_LIB: kernel experiments

c_f32p = C.c_void_p      # device pointers travel as integers
c_u8p = C.c_void_p
c_i64p = C.c_void_p
c_i32p = C.c_void_p
c_f64p = C.c_void_p


class GenWeights(C.Structure):
    """struct ddg_gen_weights (include/ddgani.h)."""
    _fields_ = [("in_dim", C.c_int32), ("enc_dim", C.c_int32), ("lat_dim", C.c_int32), ("n_steps", C.c_int32)] + \
               [(n, C.c_void_p) for n in ("W1", "b1", "g1", "be1", "W2", "b2", "g2", "be2",
                                          "W3", "b3", "g3", "be3", "W4", "b4", "temb")]


class DiscWeights(C.Structure):
    """struct ddg_disc_weights (include/ddgani.h)."""
    _fields_ = [("in_dim", C.c_int32), ("lat_dim", C.c_int32), ("out_dim", C.c_int32), ("n_steps", C.c_int32)] + \
               [(n, C.c_void_p) for n in ("W1", "b1", "W2", "b2", "W3", "b3", "W4", "b4", "e1", "e2", "e3")]


class TrainBatch(C.Structure):
    """struct ddg_train_batch (include/ddgani.h)."""
    _fields_ = [(n, C.c_void_p) for n in ("x0", "M", "m", "t", "eps", "drop0", "drop1", "drop2")] + \
               [("drop_p", C.c_float), ("seed", C.c_uint64), ("stream_id", C.c_uint64), ("rows", C.c_int64)]


class GenGrads(C.Structure):
    """struct ddg_gen_grads (include/ddgani.h)."""
    _fields_ = [(n, C.c_void_p) for n in ("W1", "b1", "g1", "be1", "W2", "b2", "g2", "be2",
                                          "W3", "b3", "g3", "be3", "W4", "b4", "temb")]


class DiscGrads(C.Structure):
    """struct ddg_disc_grads (include/ddgani.h)."""
    _fields_ = [(n, C.c_void_p) for n in ("W1", "b1", "W2", "b2", "W3", "b3", "W4", "b4", "e1", "e2", "e3")]


class FdWeights(C.Structure):
    """struct ddg_fd_weights (include/ddgani.h)."""
    _fields_ = [("in_dim", C.c_int32), ("out_dim", C.c_int32)] + [(n, C.c_void_p) for n in ("W1", "b1", "W3", "b3")]


class DplainWeights(C.Structure):
    """struct ddg_dplain_weights (include/ddgani.h)."""
    _fields_ = [("in_dim", C.c_int32), ("out_dim", C.c_int32)] + [(n, C.c_void_p) for n in ("W1", "b1", "W2", "b2", "W3", "b3")]


ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p)   # ddg_allreduce_fn


class Dp(C.Structure):
    """struct ddg_dp (include/ddgani.h)."""
    _fields_ = [("global_rows", C.c_int64), ("allreduce", ALLREDUCE_FN), ("ctx", C.c_void_p)]


# name -> (restype, argtypes); every symbol include/ddgani.h declares
SIGNATURES = {
    "ddg_version": (C.c_int, []),
    "ddg_error_string": (C.c_char_p, [C.c_int]),
    "ddg_schedule": (C.c_int, [C.c_int, C.POINTER(C.c_float)]),
    "ddg_forward_noise": (C.c_int, [c_f32p, c_f32p, c_i64p, C.c_int, c_f32p, C.c_int, C.c_uint64, C.c_uint64,
                                    c_f32p, c_f32p, C.c_int64, C.c_int, C.c_void_p]),
    "ddg_reverse_step": (C.c_int, [c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, C.POINTER(C.c_float), C.c_int, C.c_int,
                                   C.c_uint64, C.c_uint64, c_f32p, C.c_int64, C.c_int, C.c_void_p]),
    "ddg_head_reverse_step": (C.c_int, [c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, C.POINTER(C.c_float), C.c_int, C.c_int,
                                        C.c_uint64, C.c_uint64, c_i32p, c_u8p, C.c_int, c_f32p, C.c_int64, C.c_int,
                                        C.c_void_p]),
    "ddg_head": (C.c_int, [c_f32p, c_i32p, c_u8p, C.c_int, c_f32p, C.c_int64, C.c_int, C.c_void_p]),
    "ddg_mask_merge": (C.c_int, [c_f32p, c_f32p, c_f32p, c_i32p, c_u8p, C.c_int, c_f32p, C.c_int64, C.c_int,
                                 C.c_void_p]),
    "ddg_linear": (C.c_int, [c_f32p, c_f32p, C.c_int, c_f32p, c_f32p, C.c_int, c_f32p, c_i64p, C.c_int, c_f32p,
                             c_f32p, c_u8p, C.c_float, C.c_uint64, C.c_uint64, c_f32p, c_f64p, C.c_int64, C.c_int,
                             C.c_int, C.c_void_p]),
    "ddg_bn_finalize": (C.c_int, [c_f64p, c_f32p, c_f32p, C.c_double, C.c_float, c_f32p, c_f32p, C.c_int,
                                  C.c_void_p]),
    "ddg_denoiser_workspace_bytes": (C.c_size_t, [C.c_int64, C.c_int, C.c_int, C.c_int]),
    "ddg_denoiser_fwd": (C.c_int, [C.POINTER(GenWeights), c_f32p, c_i64p, C.c_int, c_u8p, c_u8p, c_u8p, C.c_float,
                                   C.c_uint64, C.c_uint64, C.c_float, c_i32p, c_u8p, C.c_int, C.c_int, c_f32p,
                                   C.c_int64, C.c_void_p, C.c_size_t, C.c_void_p]),
    "ddg_denoiser_fwd_static": (C.c_int, [C.POINTER(GenWeights), c_f32p, c_i64p, C.c_int, c_u8p, c_u8p, c_u8p, C.c_float,
                                          C.c_uint64, C.c_uint64, C.c_float, c_i32p, c_u8p, C.c_int, C.c_int, c_f32p,
                                          C.c_int64, C.c_void_p]),
    "ddg_disc_workspace_bytes": (C.c_size_t, [C.c_int64, C.c_int]),
    "ddg_disc_fwd": (C.c_int, [C.POINTER(DiscWeights), c_f32p, c_f32p, c_i64p, C.c_int, c_f32p, C.c_int64,
                               C.c_void_p, C.c_size_t, C.c_void_p]),
    "ddg_train_workspace_bytes": (C.c_size_t, [C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
    "ddg_train_gen_grads": (C.c_int, [C.POINTER(GenWeights), C.POINTER(DiscWeights), C.POINTER(TrainBatch), c_f32p,
                                      C.c_int, c_i32p, c_u8p, C.c_int, C.c_float, C.c_float, C.c_float,
                                      C.POINTER(GenGrads), c_f32p, c_f32p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "ddg_train_disc_grads": (C.c_int, [C.POINTER(GenWeights), C.POINTER(DiscWeights), C.POINTER(TrainBatch), c_f32p,
                                       C.c_int, c_i32p, c_u8p, C.c_int, C.c_float, C.c_float,
                                       C.POINTER(DiscGrads), c_f32p, c_f32p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "ddg_train_gen_grads_dp": (C.c_int, [C.POINTER(GenWeights), C.POINTER(DiscWeights), C.POINTER(TrainBatch), c_f32p,
                                         C.c_int, c_i32p, c_u8p, C.c_int, C.c_float, C.c_float, C.c_float,
                                         C.POINTER(GenGrads), c_f32p, c_f32p, C.c_void_p, C.c_size_t, C.c_void_p,
                                         C.POINTER(Dp)]),
    "ddg_train_disc_grads_dp": (C.c_int, [C.POINTER(GenWeights), C.POINTER(DiscWeights), C.POINTER(TrainBatch), c_f32p,
                                          C.c_int, c_i32p, c_u8p, C.c_int, C.c_float, C.c_float,
                                          C.POINTER(DiscGrads), c_f32p, c_f32p, C.c_void_p, C.c_size_t, C.c_void_p,
                                          C.POINTER(Dp)]),
    "ddg_set_stream_offset": (C.c_int, [C.c_void_p]),
    "ddg_gather_batch": (C.c_int, [c_f32p, c_f32p, c_f32p, c_i64p, C.c_int64, C.c_int, C.c_int, C.c_int, c_f32p, c_f32p,
                                   c_f32p, c_i64p, c_i64p, C.c_void_p]),
    "ddg_step_end": (C.c_int, [c_i64p, c_i64p, C.c_void_p]),
    "ddg_adam_step": (C.c_int, [C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_void_p),
                                C.POINTER(C.c_void_p), C.POINTER(C.c_int64), C.c_int, C.c_double, C.c_double,
                                C.c_double, C.c_double, C.c_double, C.c_void_p]),
    "ddg_attn_workspace_bytes": (C.c_size_t, [C.c_int64, C.c_int64, C.c_int, C.c_int, C.c_int64]),
    "ddg_attn_topk_fill": (C.c_int, [c_f32p, c_u8p, C.c_int64, c_f32p, c_u8p, c_f32p, C.POINTER(C.c_int32),
                                     C.c_int64, C.c_int64, C.c_int, C.c_int, C.c_int64, C.c_int, c_f32p, c_f32p,
                                     c_i32p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "ddg_attn_prepare": (C.c_int, [c_f32p, c_u8p, C.c_int64, c_f32p, c_u8p, c_f32p, C.POINTER(C.c_int32), C.c_int64,
                                   C.c_int64, C.c_int, C.c_int, C.c_int64, C.c_int, C.c_void_p, C.c_size_t,
                                   C.c_void_p]),
    "ddg_attn_select": (C.c_int, [c_f32p, C.c_int64, C.c_int64, C.c_int, C.c_int, C.c_int64, C.c_int, C.c_int, c_f32p,
                                  C.c_void_p, C.c_size_t, C.c_void_p]),
    "ddg_attn_fallback_pairs": (C.c_int, [C.c_int64, C.c_int64, C.c_int, C.c_int, C.c_int64, C.c_int, C.c_void_p, C.c_void_p,
                                          C.c_size_t, C.c_void_p]),
    "ddg_attn_set_key_chunk": (C.c_int, [C.c_int64]),
    "ddg_attn_fill": (C.c_int, [C.POINTER(C.c_int32), C.c_int64, C.c_int64, C.c_int, C.c_int, C.c_int64, C.c_int,
                                C.c_int, c_f32p, c_f32p, c_i32p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "ddg_fd_workspace_bytes": (C.c_size_t, [C.c_int64, C.c_int, C.c_int]),
    "ddg_fd_train_epoch": (C.c_int, [C.POINTER(FdWeights), c_f32p, c_i32p, c_i64p, C.c_int64, C.c_int, C.c_float,
                                     C.c_void_p, C.c_size_t, C.c_void_p]),
    "ddg_fd_eval": (C.c_int, [C.POINTER(FdWeights), c_f32p, c_i32p, C.c_int64, c_i32p, c_f32p, C.c_void_p, C.c_size_t,
                              C.c_int64, C.c_void_p]),
    "ddg_contingency": (C.c_int, [C.c_void_p, C.c_int64, C.c_int, c_i32p, c_i32p, c_i32p, c_i64p, C.c_int64, C.c_int64,
                                  C.c_void_p, C.c_void_p]),
    "ddg_score_table": (C.c_int, [c_f32p, c_f64p, c_u8p, c_i32p, c_u8p, c_f64p, c_f64p, c_u8p, c_f64p, c_f64p, c_f64p, c_f64p,
                                  C.c_int64, C.c_int, C.c_int, C.c_void_p]),
    "ddg_encode_table": (C.c_int, [c_f64p, c_i32p, c_u8p, c_u8p, c_f64p, c_f64p, c_f32p, C.c_int64, C.c_int, C.c_int,
                                   C.c_void_p]),
    "ddg_decode_table": (C.c_int, [c_f32p, c_u8p, c_f64p, c_i32p, c_u8p, c_f64p, c_f64p, c_u8p, c_f64p, c_f64p, C.c_int64,
                                   C.c_int, C.c_int, C.c_void_p]),
    "ddg_dplain_workspace_bytes": (C.c_size_t, [C.c_int64, C.c_int, C.c_int]),
    "ddg_dplain_grads": (C.c_int, [C.POINTER(DplainWeights), c_f32p, c_f32p, c_u8p, c_u8p, C.c_float, C.c_uint64, C.c_uint64,
                                   C.c_int64, C.POINTER(DplainWeights), C.c_void_p, C.c_size_t, C.c_void_p]),
    "ddg_dplain_fwd": (C.c_int, [C.POINTER(DplainWeights), c_f32p, c_u8p, c_u8p, C.c_float, C.c_uint64, C.c_uint64, C.c_int64,
                                 c_f32p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "ddg_learner_workspace_bytes": (C.c_size_t, [C.c_int, C.c_int]),
    "ddg_learner_train": (C.c_int, [c_f32p, c_i32p, C.c_int64, c_f32p, c_i32p, C.c_int64, C.c_int, C.c_int, C.c_int,
                                    C.c_float, c_f32p, c_f32p, c_f32p, c_i32p, C.c_void_p, C.c_size_t, C.c_void_p]),
}

_lib = None


class DdgError(RuntimeError):
    pass


def lib() -> C.CDLL:
    """Load the HIP library (once).  Raises if it has not been built: no fallback exists."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise DdgError(f"{LIB_PATH} not found: build it with `make -C ddgani_amd/csrc` "
                           "(or __graft_entry__.build()); ddgani_amd has no CPU fallback")
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def check(rc: int, what: str) -> None:
    if rc != 0:
        raise DdgError(f"{what} failed: {lib().ddg_error_string(rc).decode()} ({rc})")
