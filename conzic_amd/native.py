"""ctypes binding of libconzic_hip.so (include/conzic_hip.h).

The shared library is the product; this file only declares its C ABI.  There is no CPU
fallback: if the library or a GPU is missing, calls fail loudly (`NativeError`).
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

_HERE = os.path.dirname(os.path.abspath(__file__))
# CZC_LIB_PATH: a differently-built library for the kernel tools (`make EXPERIMENTS=1 LIB=...`: timing-ablation kernels)
LIB_PATH = os.environ.get("CZC_LIB_PATH") or os.path.join(_HERE, "lib", "libconzic_hip.so")
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "conzic_hip.h")
# kernel-level parity hooks + GEMM microbenchmark: a second library over the product one, for tests/ and tools/ only
# (csrc/Makefile puts it next to the product library it was linked against: libconzic_hip<tag>.so -> libconzic_hip<tag>_test.so)
TEST_LIB_PATH = os.environ.get("CZC_TEST_LIB_PATH") or (LIB_PATH[:-3] + "_test.so" if LIB_PATH.endswith(".so")
                                                        else os.path.join(os.path.dirname(LIB_PATH), "libconzic_hip_test.so"))
TEST_HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "conzic_hip_test.h")

PREC_BF16 = 0
PREC_F32 = 1
PREC_ALL_BF16 = 2
PREC_SPLIT = 3
PREC_FP16 = 4
PREC_REFINE = 5
CLIP_MAX_LEN = 77


ERR_ARG, ERR_HIP, ERR_STATE, ERR_OVERFLOW = 1, 2, 3, 4   # include/conzic_hip.h CZC_ERR_*


class NativeError(RuntimeError):
    """A non-zero status of the C ABI; `code` is that status (CZC_ERR_*), None where no call was made."""

    def __init__(self, msg, code=None):
        super().__init__(msg)
        self.code = code


class Config(C.Structure):
    _fields_ = [
        ("bert_vocab", C.c_int32), ("bert_hidden", C.c_int32), ("bert_layers", C.c_int32), ("bert_heads", C.c_int32),
        ("bert_inter", C.c_int32), ("bert_max_pos", C.c_int32), ("bert_eps", C.c_float),
        ("clip_vocab", C.c_int32), ("clip_hidden", C.c_int32), ("clip_layers", C.c_int32), ("clip_heads", C.c_int32),
        ("clip_inter", C.c_int32), ("clip_max_pos", C.c_int32), ("clip_proj", C.c_int32), ("clip_eps", C.c_float),
        ("clip_bos_id", C.c_int32), ("clip_eos_id", C.c_int32),
        ("vis_hidden", C.c_int32), ("vis_layers", C.c_int32), ("vis_heads", C.c_int32), ("vis_inter", C.c_int32),
        ("vis_image", C.c_int32), ("vis_patch", C.c_int32),
        ("pad_id", C.c_int32), ("unk_id", C.c_int32), ("cls_id", C.c_int32), ("sep_id", C.c_int32),
        ("mask_id", C.c_int32), ("dot_id", C.c_int32),
        ("precision", C.c_int32),
    ]


class BridgeTables(C.Structure):
    _fields_ = [
        ("bert_vocab", C.c_int32),
        ("piece_off", C.c_void_p), ("piece_bytes", C.c_void_p), ("piece_class", C.c_void_p), ("piece_flags", C.c_void_p),
        ("clip_vocab", C.c_int32),
        ("byte_sym", C.c_void_p), ("byte_sym_eow", C.c_void_p),
        ("n_merges", C.c_int32),
        ("merge_left", C.c_void_p), ("merge_right", C.c_void_p), ("merge_out", C.c_void_p),
        ("bos_id", C.c_int32), ("eos_id", C.c_int32),
    ]


class Hyper(C.Structure):
    _fields_ = [("alpha", C.c_float), ("beta", C.c_float), ("gamma", C.c_float), ("temperature", C.c_float),
                ("control", C.c_int32), ("negative", C.c_int32)]


STEP_OUT_FIELDS = ["probs", "idxs", "cand_ids", "clip_ids", "clip_len", "clip_score", "clip_ref", "senti_raw",
                   "repeats", "final_score", "best", "best_cos", "logits"]


class StepOut(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in STEP_OUT_FIELDS]


# czc_control_fn (include/conzic_hip.h): host scorer of the K candidate sentences of every image, called once per step
CONTROL_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.c_int, C.c_int, C.c_int,
                         C.c_int, C.POINTER(C.c_float))

# every entry point include/conzic_hip.h declares: name -> (restype, argtypes)
_P = C.c_void_p
_I = C.c_int
SIGNATURES = {
    "czc_create": (_I, [C.POINTER(Config), _I, C.POINTER(_P)]),
    "czc_destroy": (_I, [_P]),
    "czc_last_error": (C.c_char_p, [_P]),
    "czc_version": (_I, []),
    "czc_load_tensor": (_I, [_P, C.c_char_p, _I, _I, C.POINTER(C.c_int64), _P]),
    "czc_finalize_weights": (_I, [_P]),
    "czc_set_token_mask": (_I, [_P, _P, _I]),
    "czc_set_bridge": (_I, [_P, C.POINTER(BridgeTables)]),
    "czc_set_lexicon": (_I, [_P, _P, _I]),
    "czc_set_lexicon_pos": (_I, [_P, _P, _P, _I]),
    "czc_set_pos": (_I, [_P, _P, _I, _P, _I]),
    "czc_set_control_callback": (_I, [_P, _P, _P]),
    "czc_encode_images": (_I, [_P, _P, _I, _P]),
    "czc_preprocess_u8": (_I, [_P, _P, _I, _I, _P, _P, _I, _P]),
    "czc_encode_staged": (_I, [_P, _I, _P]),
    "czc_set_image_embeds": (_I, [_P, _P, _I]),
    "czc_encode_text": (_I, [_P, _P, _P, _I, _P]),
    "czc_similarity": (_I, [_P, _P, _P, _I, _I, _P, _P]),
    "czc_step": (_I, [_P, _P, _I, _I, _I, _I, _I, _I, C.POINTER(Hyper), C.POINTER(StepOut)]),
    "czc_generate": (_I, [_P, _I, _I, _I, _I, _P, _I, _I, _P, _P, _I, C.POINTER(Hyper), _P, _P]),
    "czc_set_option": (_I, [_P, C.c_char_p, _I]),
    "czc_get_option": (_I, [_P, C.c_char_p, C.POINTER(_I)]),
    "czc_profile_enable": (_I, [_P, _I]),
    "czc_profile_reset": (_I, [_P]),
    "czc_profile_get": (_I, [_P, C.c_char_p, C.POINTER(C.c_double), C.POINTER(C.c_int64), C.POINTER(C.c_double)]),
    "czc_profile_intervals": (_I, [_P, _P, C.c_char_p, _P, _P, _I, C.POINTER(C.c_int)]),
    "czc_replicate": (_I, [_P, C.POINTER(C.c_void_p)]),
    "czc_sync": (_I, [_P]),
    "czc_stats": (_I, [_P, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "czc_refine_stats": (_I, [_P, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "czc_dedup_stats": (_I, [_P, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "czc_refine_guard": (_I, [_P, _I, C.POINTER(C.c_float), C.POINTER(C.c_int64)]),
    "czc_refine_gate_stats": (_I, [_P, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
}
# exported, but not part of the boundary and not in the public header: the hook library's door into this one (declared in
# csrc/kernels.h; nothing in conzic_amd/ calls it)
PRIVATE_SIGNATURES = {"czc_internal_hooks": (_P, [_I])}
# every entry point include/conzic_hip_test.h declares (libconzic_hip_test.so)
TEST_SIGNATURES = {
    "czc_test_gemm": (_I, [_I, _I, _I, _I, _P, _P, _P, _P, _I, _P]),
    "czc_test_gemm_x16": (_I, [_I, _I, _I, _I, _P, _P, _P, _P, _P, _P]),
    "czc_test_ln_fold_gemm": (_I, [_I, _I, _I, _P, _P, _P, _P, _P, _P, C.c_float, _I, _P, _P]),
    "czc_test_layernorm_x16": (_I, [_I, _I, _P, _P, _P, C.c_float, _P]),
    "czc_test_gemm_rowln": (_I, [_I, _I, _I, _P, _P, _P, _P, _P, _P, C.c_float, _P, _P]),
    "czc_bench_gemm": (_I, [_I, _I, _I, _I, _I, _I, _I, _I, C.POINTER(C.c_double)]),
    "czc_test_set_option": (_I, [C.c_char_p, _I]),
    "czc_test_layernorm": (_I, [_I, _I, _I, _P, _P, _P, C.c_float, _P]),
    "czc_test_attention": (_I, [_I, _I, _P, _I, _I, C.c_float, _P, _P]),
    "czc_test_topk": (_I, [_I, _I, _I, _P, _P, C.c_float, _I, _I, _P, _P, _P]),
    "czc_test_bridge": (_I, [C.POINTER(BridgeTables), C.POINTER(Config), _I, _I, _P, _P, _P]),
    "czc_test_combine": (_I, [_I, _I, _I, _P, _P, C.c_float, _P, _P, _P, C.POINTER(Hyper), _P, _P, _P, _P]),
}

_lib: Optional[C.CDLL] = None
_test_lib: Optional[C.CDLL] = None


def load() -> C.CDLL:
    """dlopen the in-tree library and type every symbol.  Raises NativeError when it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise NativeError(f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                          "(there is no CPU fallback)")
    lib = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    for name, (res, args) in list(SIGNATURES.items()) + list(PRIVATE_SIGNATURES.items()):
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def load_test() -> C.CDLL:
    """dlopen the kernel-level hook library (tests / tools only) on top of the product library."""
    global _test_lib
    if _test_lib is not None:
        return _test_lib
    load()  # the hook library is linked against the product library's C ABI (czc_internal_hooks)
    if not os.path.exists(TEST_LIB_PATH):
        raise NativeError(f"{TEST_LIB_PATH} not found: `make -C conzic_amd/csrc` builds it next to the product library")
    lib = C.CDLL(TEST_LIB_PATH, mode=C.RTLD_GLOBAL)
    for name, (res, args) in TEST_SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _test_lib = lib
    return lib


def check(rc: int, handle=None, what: str = ""):
    if rc != 0:
        lib = load()
        msg = lib.czc_last_error(handle)
        raise NativeError(f"{what} failed (status {rc}): {msg.decode() if msg else '?'}", code=rc)
