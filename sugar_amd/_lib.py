"""
ctypes binding of libsugar_amd.so (include/sugar_amd.h).  No torch types cross the boundary: callers pass
``tensor.data_ptr()`` integers and the raw HIP stream handle.

The product path has NO fallback: if the library is missing or a call fails this module raises.
"""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libsugar_amd.so")

SUGAR_F32, SUGAR_BF16 = 0, 1
MAX_DEPTH, RES2_SCALE = 8, 8
EPI_RELU, EPI_AFFINE, EPI_TANH = 1, 2, 4

_fp = C.POINTER(C.c_float)


class SugarBN(C.Structure):
    _fields_ = [("weight", _fp), ("bias", _fp), ("running_mean", _fp), ("running_var", _fp)]


class SugarBlock(C.Structure):
    _fields_ = [("mid_channels", C.c_int), ("kernel", C.c_int),
                ("first_w", _fp), ("first_bn", SugarBN),
                ("res_w", _fp * (RES2_SCALE - 1)), ("res_bn", SugarBN * (RES2_SCALE - 1)),
                ("last_w", _fp), ("last_bn", SugarBN),
                ("se_reduce_w", _fp), ("se_reduce_b", _fp), ("se_expand_w", _fp), ("se_expand_b", _fp)]


class SugarSubnetDesc(C.Structure):
    _fields_ = [("in_feats", C.c_int), ("embed_dim", C.c_int), ("depth", C.c_int), ("c0", C.c_int),
                ("k0", C.c_int), ("ccat", C.c_int), ("bottleneck", C.c_int),
                ("layer1_w", _fp), ("layer1_bn", SugarBN),
                ("blocks", SugarBlock * MAX_DEPTH),
                ("catconv_w", _fp), ("pool1_w", _fp), ("pool2_w", _fp), ("bnpool", SugarBN),
                ("linear_w", _fp), ("linear_b", _fp), ("bnembed", SugarBN)]


# every symbol include/sugar_amd.h declares: (restype, argtypes)
_vp, _i, _i64 = C.c_void_p, C.c_int, C.c_int64
SYMBOLS = {
    "sugar_last_error": (C.c_char_p, []),
    "sugar_version": (C.c_char_p, []),
    "sugar_subnet_create": (_i, [C.POINTER(SugarSubnetDesc), _i, _i, C.POINTER(_vp)]),
    "sugar_subnet_destroy": (_i, [_vp]),
    "sugar_workspace_bytes": (_i64, [_vp, _i, _i64]),
    "sugar_fbank_scratch_bytes": (_i64, [_i, _i64]),
    "sugar_fbank_fwd": (_i, [_vp, _vp, _vp, _vp, _i, _i64, _i, _vp, _vp, _i, _vp]),
    "sugar_fbank_fwd_noalloc": (_i, [_vp, _vp, _vp, _vp, _i, _i64, _i, _vp, _i, _vp]),
    "sugar_tdnn_fwd": (_i, [_vp, _vp, _vp, _i, _i64, _i, _vp, _i64, _vp, _i, _vp]),
    "sugar_embed_fwd": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i64, _i, _vp, _i64, _vp, _i, _vp]),
    "sugar_tdnn_tap": (_i, [_vp, _vp, _vp, _i, _i64, _i, _vp, _i64, _i, _vp, _i64, _vp, _vp]),
    "sugar_conv1d_fwd": (_i, [_fp, C.POINTER(SugarBN), _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _i, _i64, _i, _i, _vp]),
    "sugar_l2norm": (_i, [_vp, _i64, _i, _vp]),
    "sugar_profile_enable": (_i, [C.c_uint]),
    "sugar_profile_read": (_i, [_i, C.POINTER(_i), C.POINTER(C.c_double)]),
    "sugar_profile_stage_name": (C.c_char_p, [_i]),
    "sugar_cosine_trials": (_i, [_vp, _i, _vp, _i, _vp, _vp, _i64, _vp, _vp]),
    "sugar_norm2_trials": (_i, [_vp, _i, _vp, _i, _vp, _vp, _i64, _vp, _vp]),
    "sugar_pack_segments": (_i, [_vp, _vp, _i, _vp, _i]),
    "sugar_wav_probe": (_i, [_vp, _i, _i, _vp, _vp, _i]),
    "sugar_wav_load": (_i, [_vp, _vp, _vp, _i, _vp, _vp, _vp, _i, _vp, _i]),
    "sugar_cohort_scores": (_i, [_vp, _i, _vp, _i, _vp, _i, _vp, _vp, _vp]),
    "sugar_topk_stats": (_i, [_vp, _i64, _i, _i, C.c_float, _vp, _vp, _vp]),
    "sugar_snorm_trials": (_i, [_vp, _vp, _vp, _i64, _vp, _vp, C.c_float, _vp, _vp]),
    "sugar_tdnn_recalibrate": (_i, [_vp, _vp, _vp, _i, _i64, _i, _vp, _i64, C.c_float, _vp, _vp]),
    "sugar_embed_recalibrate": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i64, _i, _vp, _i64, C.c_float, _vp, _vp]),
    "sugar_subnet_bn_count": (_i, [_vp]),
    "sugar_subnet_bn_read": (_i, [_vp, _i, _fp, _fp, _i]),
    "sugar_subnet_bn_reset": (_i, [_vp]),
    "sugar_subnet_bn_commit": (_i, [_vp]),
    "sugar_tune_set": (_i, [C.c_char_p, _i64]),
    "sugar_tune_get": (_i, [C.c_char_p, C.POINTER(_i64)]),
}

_lib = None


class SugarError(RuntimeError):
    pass


def lib():
    """Load the library (once).  Raises if it has not been built -- there is no CPU fallback."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise SugarError("%s not found: build it with `python -m sugar_amd.build` (hipcc, gfx950). "
                             "sugar_amd has no CPU or PyTorch fallback path." % LIB_PATH)
        handle = C.CDLL(LIB_PATH)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(handle, name)     # AttributeError if the symbol is not exported
            fn.restype = res
            fn.argtypes = args
        _lib = handle
    return _lib


def check(rc, what=""):
    """Turn a negative return code into the reference's exception convention (ValueError / RuntimeError)."""
    if rc is not None and rc < 0:
        msg = lib().sugar_last_error().decode(errors="replace")
        if rc == -1:
            raise ValueError("%s: %s" % (what or "sugar_amd", msg))
        raise SugarError("%s: %s (code %d)" % (what or "sugar_amd", msg, rc))
    return rc


class tuned:
    """Context manager: set library tunables (include/sugar_amd.h, sugar_tune_set) and restore them on exit."""

    def __init__(self, **kw):
        self.kw, self.old = kw, {}

    def __enter__(self):
        for k, v in self.kw.items():
            cur = _i64(0)
            check(lib().sugar_tune_get(k.encode(), C.byref(cur)), "sugar_tune_get")
            self.old[k] = cur.value
            check(lib().sugar_tune_set(k.encode(), int(v)), "sugar_tune_set")
        return self

    def __exit__(self, *exc):
        for k, v in self.old.items():
            check(lib().sugar_tune_set(k.encode(), v), "sugar_tune_set")
        return False


STAGE_COUNT = 12
STAGE_CATCONV = 7


def profile_enable(mask):
    check(lib().sugar_profile_enable(mask), "sugar_profile_enable")


def profile_read():
    """{stage name: (brackets recorded, total ms)} for every stage with at least one record."""
    out = {}
    for st in range(STAGE_COUNT):
        n, ms = C.c_int(0), C.c_double(0.0)
        check(lib().sugar_profile_read(st, C.byref(n), C.byref(ms)), "sugar_profile_read")
        if n.value:
            out[lib().sugar_profile_stage_name(st).decode()] = (n.value, ms.value)
    return out


def fptr(t):
    """float* of a contiguous CPU float32 tensor (the tensor must outlive the call)."""
    assert t.device.type == "cpu" and t.dtype.is_floating_point and t.element_size() == 4 and t.is_contiguous()
    return C.cast(t.data_ptr(), _fp)
