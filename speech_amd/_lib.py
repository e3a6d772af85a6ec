"""ctypes binding of libspeech_hip.so (the C ABI declared in include/speech_hip.h).

There is no CPU fallback: if the shared library is missing, importing a kernel raises.
Build it with ``python -m speech_amd.build`` (hipcc, gfx950).
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libspeech_hip.so")

SP_F32, SP_F16, SP_BF16 = 0, 1, 2
SUMSQ_WORDS = 1025   # SP_SUMSQ_WORDS of include/speech_hip.h
ADAM8_BLOCK = 2048   # SP_ADAM8_BLOCK
CONV_WGRAD_WS_WORDS = 512 * 15 * 32 * 32   # SP_CONV_WGRAD_WS_WORDS: one slab per persistent workgroup
BN_SUMS_WORDS = 8 * 64 + 1024 * 64 + 64   # SP_BN_SUMS_WORDS: [8][64] rows, per-workgroup partials, their count

_c = ctypes
_vp, _i, _i64, _f, _sz = _c.c_void_p, _c.c_int, _c.c_int64, _c.c_float, _c.c_size_t

# name -> (restype, argtypes); mirrors include/speech_hip.h one to one
SIGNATURES = {
    "sp_last_error": (_c.c_char_p, []),
    "sp_version": (_i, []),
    "sp_melspec_plan_bytes": (_sz, []),
    "sp_melspec_plan": (_i, [_vp, _vp, _vp]),
    "sp_melspec_fwd": (_i, [_vp, _vp, _i, _i, _vp, _vp, _i, _vp, _vp]),
    "sp_ctc_workspace_bytes": (_sz, [_i, _i, _i, _i]),
    "sp_ctc_fwd_bwd": (_i, [_vp, _i64, _i64, _vp, _i, _vp, _vp, _i, _i, _i, _i, _i, _f, _vp, _vp,
                            _vp, _i, _vp, _sz, _vp]),
    "sp_ctc_noalloc_max_bytes": (_sz, []),
    "sp_ctc_fwd_bwd_noalloc": (_i, [_vp, _i64, _i64, _vp, _i, _vp, _vp, _i, _i, _i, _i, _i, _f, _vp, _vp,
                                    _vp, _i, _vp]),
    "sp_greedy_decode": (_i, [_vp, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp]),
    "sp_edit_distance": (_i, [_vp, _i, _vp, _i]),
    "sp_beam_candidates": (_i, [_vp, _i, _i, _i, _i, _f, _vp, _i, _vp, _vp, _vp, _vp]),
    "sp_ngram_load_arpa": (_i, [_c.c_char_p, _f, _f, _f, _i, _c.POINTER(_vp)]),
    "sp_ngram_free": (None, [_vp]),
    "sp_ngram_order": (_i, [_vp]),
    "sp_ngram_score_sentence": (_i, [_vp, _c.POINTER(_c.c_char_p), _i, _i, _i, _c.POINTER(_c.c_double)]),
    "sp_ctc_beam_decode": (_i, [_vp, _vp, _vp, _i, _i, _i, _vp, _vp, _i, _i, _i, _vp, _i, _f, _i, _i, _i, _vp, _vp, _vp]),
    "sp_sru_fwd": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp, _vp]),
    "sp_sru_bwd": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp,
                        _vp]),
    "sp_sumsq_f32": (_i, [_vp, _i64, _vp, _vp]),
    "sp_adam_step_f32": (_i, [_vp, _vp, _vp, _vp, _i64, _f, _f, _f, _f, _f, _i, _i, _vp, _f, _f, _vp]),
    "sp_adam8_step_f32": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _f, _f, _f, _f, _f, _i, _i, _vp, _f, _f, _vp]),
    "sp_conv_pack_w": (_i, [_vp, _i, _vp, _vp]),
    "sp_conv_pack_w_batched": (_i, [_vp, _i, _i, _vp, _vp]),
    "sp_conv3x5": (_i, [_vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp]),
    "sp_conv3x5_wgrad": (_i, [_vp, _vp, _i, _vp, _vp, _vp, _i, _i, _i, _vp, _vp, _vp]),
    "sp_conv_stem_fwd": (_i, [_vp, _vp, _vp, _i, _i, _i, _vp, _vp]),
    "sp_conv_stem_bwd": (_i, [_vp, _vp, _i, _i, _i, _vp, _vp, _vp]),
    "sp_bn_stats": (_i, [_vp, _i64, _vp, _vp]),
    "sp_bn_bwd_reduce": (_i, [_vp, _vp, _i64, _vp, _vp]),
    "sp_bn_finalize_fwd": (_i, [_vp, _i64, _vp, _vp, _vp, _vp, _f, _f, _i, _vp, _vp, _vp]),
    "sp_bn_finalize_bwd": (_i, [_vp, _vp, _vp, _i64, _vp, _vp, _vp, _vp]),
    "sp_block_out": (_i, [_vp, _vp, _vp, _i, _i, _i, _vp, _i64, _i64, _i64, _vp]),
    "sp_block_bwd_in": (_i, [_vp, _vp, _i64, _i64, _i64, _vp, _i, _i, _i, _vp, _vp, _vp]),
    "sp_conv2d_f32": (_i, [_vp, _i64, _i64, _i64, _i, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _i, _i, _i, _i, _i, _i, _i,
                           _vp, _i64, _i64, _i64, _vp]),
    "sp_sru_pack_w": (_i, [_vp, _i, _i, _i, _i, _i, _vp, _vp]),
    "sp_sru_pack_wt": (_i, [_vp, _i, _i, _i, _i, _i, _vp, _i64, _vp, _vp]),
    "sp_sru_pack_wt_batched": (_i, [_vp, _i, _i, _i, _i, _i, _i, _vp, _i64, _vp, _vp]),
    "sp_sru_unpack_dw": (_i, [_vp, _i, _i, _i, _i, _i, _i, _i, _vp, _vp]),
    "sp_gemm": (_i, [_vp, _vp, _vp, _i, _i, _i, _i64, _i64, _i64, _i64, _i64, _i, _i, _i, _i, _i64, _vp]),
    "sp_gemm_tn_splits": (_i, [_i, _i, _i, _i]),
    "sp_sum_slabs_f32": (_i, [_vp, _i, _i64, _i64, _vp, _i, _vp, _vp]),
    "sp_linear_pack": (_i, [_vp, _i, _i, _i, _vp, _i, _vp, _vp, _vp]),
    "sp_layernorm_fwd": (_i, [_vp, _vp, _vp, _i64, _i, _f, _i, _vp, _i64, _vp, _vp, _vp]),
    "sp_layernorm_bwd": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _i, _i, _vp, _vp, _vp, _vp]),
}

_lib = None


class SpeechHipError(RuntimeError):
    pass


def lib():
    """Load (once) and return the ctypes handle.  Raises if the library was not built."""
    global _lib
    if _lib is None:
        # torch ships its own HIP runtime; it must be the one already mapped when this library's
        # libamdhip64 dependency is resolved, or the process ends up with two runtimes.
        import torch  # noqa: F401
        if not os.path.exists(LIB_PATH):
            raise SpeechHipError(
                f"{LIB_PATH} not found: build it with `python -m speech_amd.build` "
                "(there is no CPU fallback for the HIP kernels)")
        h = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(h, name)
            fn.restype = res
            fn.argtypes = args
        _lib = h
    return _lib


def check(rc, what=""):
    if rc != 0:
        msg = lib().sp_last_error()
        raise SpeechHipError(f"{what} failed ({rc}): {msg.decode() if msg else ''}")


def stream_ptr():
    import torch
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


def dtype_code(t):
    import torch
    return {torch.float32: SP_F32, torch.float16: SP_F16, torch.bfloat16: SP_BF16}[t.dtype]
