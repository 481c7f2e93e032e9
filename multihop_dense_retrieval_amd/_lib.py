"""ctypes binding of libmdrhip.so (include/mdr_hip.h). No fallback: if the HIP library is missing the
import fails loudly -- the product path never routes through a CPU implementation."""
import ctypes
import os

import torch  # noqa: F401  -- must be imported first: its bundled libamdhip64.so.7 is the HIP runtime
#                              both torch and libmdrhip.so must share (same SONAME -> one runtime).

_HERE = os.path.dirname(os.path.abspath(__file__))
# MDR_LIB_PATH: measurement only -- A/B a variant build of the SAME sources (build.py --out=...); never a different backend
LIB_PATH = os.environ.get("MDR_LIB_PATH") or os.path.join(_HERE, "libmdrhip.so")

MDR_DT_F32, MDR_DT_BF16, MDR_DT_F16 = 0, 1, 2
MDR_STORE_F32X2H, MDR_STORE_BF16, MDR_STORE_F32X2H_COMPACT = 0, 1, 2


class MdrError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"libmdrhip error {code}: {msg}")
        self.code = code


class EncoderConfig(ctypes.Structure):
    _fields_ = [("vocab", ctypes.c_int), ("hidden", ctypes.c_int), ("layers", ctypes.c_int), ("heads", ctypes.c_int),
                ("ffn", ctypes.c_int), ("max_pos", ctypes.c_int), ("pad_id", ctypes.c_int), ("ln_eps", ctypes.c_float),
                ("residual_fp32", ctypes.c_int)]


class Tensor(ctypes.Structure):
    _fields_ = [("name", ctypes.c_char_p), ("data", ctypes.c_void_p), ("numel", ctypes.c_int64)]


_lib = None

_c = ctypes
_SIGNATURES = {
    "mdr_last_error": (_c.c_char_p, []),
    "mdr_version": (_c.c_char_p, []),
    "mdr_index_create": (_c.c_int, [_c.c_int, _c.c_int, _c.c_int, _c.POINTER(_c.c_void_p)]),
    "mdr_index_free": (_c.c_int, [_c.c_void_p]),
    "mdr_index_reserve": (_c.c_int, [_c.c_void_p, _c.c_int64]),
    "mdr_index_add": (_c.c_int, [_c.c_void_p, _c.c_void_p, _c.c_int64, _c.c_int, _c.c_int, _c.c_void_p]),
    "mdr_index_ntotal": (_c.c_int64, [_c.c_void_p]),
    "mdr_index_dim": (_c.c_int, [_c.c_void_p]),
    "mdr_index_stream_bytes": (_c.c_int64, [_c.c_void_p]),
    "mdr_index_queries_per_pass": (_c.c_int, [_c.c_void_p, _c.c_int, _c.c_int]),
    "mdr_index_search_workspace_bytes": (_c.c_size_t, [_c.c_void_p, _c.c_int, _c.c_int]),
    "mdr_index_search": (_c.c_int, [_c.c_void_p, _c.c_void_p, _c.c_int, _c.c_int, _c.c_void_p, _c.c_void_p, _c.c_int64,
                                    _c.c_void_p, _c.c_size_t, _c.c_void_p]),
    "mdr_index_search_telemetry": (_c.c_int, [_c.c_void_p, _c.c_int, _c.c_int, _c.c_void_p, _c.POINTER(_c.c_int64), _c.c_void_p]),
    "mdr_index_set_variant": (_c.c_int, [_c.c_void_p, _c.c_int]),
    "mdr_index_last_kernel": (_c.c_char_p, [_c.c_void_p]),
    "mdr_topk_merge": (_c.c_int, [_c.c_void_p, _c.c_void_p, _c.c_int, _c.c_int, _c.c_int, _c.c_void_p, _c.c_void_p, _c.c_void_p]),
    "mdr_topk_packed_bytes": (_c.c_size_t, [_c.c_int, _c.c_int]),
    "mdr_topk_packed_ids_offset": (_c.c_size_t, [_c.c_int, _c.c_int]),
    "mdr_topk_merge_packed": (_c.c_int, [_c.c_void_p, _c.c_int, _c.c_int, _c.c_int, _c.c_void_p, _c.c_void_p, _c.c_void_p]),
    "mdr_assemble_hop2": (_c.c_int, [_c.c_void_p, _c.c_void_p, _c.c_int, _c.c_int, _c.c_void_p, _c.c_int, _c.c_void_p, _c.c_void_p, _c.c_void_p,
                                     _c.c_int64, _c.c_void_p, _c.c_int, _c.c_int, _c.c_int, _c.c_int, _c.c_void_p, _c.c_void_p, _c.c_void_p]),
    "mdr_upload_host": (_c.c_int, [_c.c_void_p, _c.c_void_p, _c.c_size_t, _c.c_int, _c.c_void_p]),
    "mdr_encoder_create": (_c.c_int, [_c.POINTER(EncoderConfig), _c.POINTER(Tensor), _c.c_int, _c.c_int, _c.c_int, _c.c_void_p,
                                      _c.POINTER(_c.c_void_p)]),
    "mdr_encoder_free": (_c.c_int, [_c.c_void_p]),
    "mdr_encoder_workspace_bytes": (_c.c_size_t, [_c.c_void_p, _c.c_int, _c.c_int]),
    "mdr_encoder_forward": (_c.c_int, [_c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_int, _c.c_int, _c.c_void_p, _c.c_void_p,
                                       _c.c_size_t, _c.c_void_p]),
    "mdr_encoder_set_fill_hint": (_c.c_int, [_c.c_void_p, _c.c_float]),
    "mdr_test_gemm_f16": (_c.c_int, [_c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_int, _c.c_void_p, _c.c_int, _c.c_int, _c.c_void_p, _c.c_int,
                                     _c.c_int, _c.c_int, _c.c_void_p]),
}
# include/mdr_hip_measure.h: exported by measurement builds only (MDR_LIB_PATH=...); bound when present, absent from the product library
_MEASURE_SIGNATURES = {
    "mdr_test_gemm_stamps": (_c.c_int, [_c.POINTER(_c.c_uint64), _c.c_int]),
    "mdr_test_i8_stamps": (_c.c_int, [_c.POINTER(_c.c_uint64), _c.c_int]),
    "mdr_test_attn_stamps": (_c.c_int, [_c.POINTER(_c.c_uint64), _c.c_int]),
}

EXPORTED_SYMBOLS = tuple(_SIGNATURES)


def lib():
    """The loaded library; raises if it has not been built (python -m multihop_dense_retrieval_amd.build)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f"{LIB_PATH} is missing: build it with `python -m multihop_dense_retrieval_amd.build` "
                               "(hipcc --offload-arch=gfx950). There is no CPU fallback.")
        L = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in _SIGNATURES.items():
            fn = getattr(L, name)  # AttributeError if the header and the library disagree
            fn.restype, fn.argtypes = res, args
        for name, (res, args) in _MEASURE_SIGNATURES.items():
            fn = getattr(L, name, None)
            if fn is not None:
                fn.restype, fn.argtypes = res, args
        _lib = L
    return _lib


def check(code):
    if code != 0:
        raise MdrError(code, lib().mdr_last_error().decode("utf-8", "replace"))


def current_stream_ptr(device=None):
    """hipStream_t of torch's current stream as an integer for the `void* stream` arguments."""
    return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)
