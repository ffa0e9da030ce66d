"""ctypes binding of include/mkhe.h (libmkhe_hip.so).  Fails loudly when the HIP library is absent."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MKHE_LIB") or os.path.join(_HERE, "lib", "libmkhe_hip.so")      # MKHE_LIB: diagnostic builds

u64p = C.POINTER(C.c_uint64)
i32p = C.POINTER(C.c_int)
s32p = C.POINTER(C.c_int32)
vp = C.c_void_p
vpp = C.POINTER(C.c_void_p)

# name -> (restype, argtypes); mirrors include/mkhe.h declaration by declaration
SIGNATURES = {
    "mkhe_last_error": (C.c_char_p, []),
    "mkhe_device_count": (C.c_int, []),
    "mkhe_ctx_create": (C.c_int, [vpp, C.c_int, u64p, C.c_int, u64p, C.c_int, C.c_int, u64p, u64p, C.c_int]),
    "mkhe_ctx_destroy": (None, [vp]),
    "mkhe_ctx_sync": (C.c_int, [vp]),
    "mkhe_ctx_wait_for": (C.c_int, [vp, vp]),
    "mkhe_capture_begin": (C.c_int, [vp]),
    "mkhe_capture_end": (C.c_int, [vp, vpp]),
    "mkhe_graph_launch": (C.c_int, [vp, vp]),
    "mkhe_graph_destroy": (None, [vp]),
    "mkhe_ct_copy": (C.c_int, [vp, vp, vp]),
    "mkhe_ctx_alpha": (C.c_int, [vp]),
    "mkhe_ctx_beta": (C.c_int, [vp, C.c_int]),
    "mkhe_ctx_n": (C.c_int, [vp]),
    "mkhe_ctx_swk_words": (C.c_size_t, [vp]),
    "mkhe_ctx_psi": (C.c_uint64, [vp, C.c_int]),
    "mkhe_ctx_stream": (vp, [vp]),
    "mkhe_swk_create": (C.c_int, [vp, vpp]),
    "mkhe_swk_create_uninit": (C.c_int, [vp, vpp]),
    "mkhe_swk_destroy": (None, [vp, vp]),
    "mkhe_swk_upload": (C.c_int, [vp, vp, u64p]),
    "mkhe_swk_upload_limbs": (C.c_int, [vp, vp, vpp, C.c_int]),
    "mkhe_swk_download": (C.c_int, [vp, vp, u64p]),
    "mkhe_swk_devptr": (vp, [vp]),
    "mkhe_ct_create": (C.c_int, [vp, C.c_int, i32p, C.c_int, vpp]),
    "mkhe_ct_create_uninit": (C.c_int, [vp, C.c_int, i32p, C.c_int, vpp]),
    "mkhe_ct_destroy": (None, [vp, vp]),
    "mkhe_ct_upload": (C.c_int, [vp, vp, u64p]),
    "mkhe_ct_upload_poly_limbs": (C.c_int, [vp, vp, C.c_int, vpp]),
    "mkhe_ct_download": (C.c_int, [vp, vp, u64p]),
    "mkhe_ct_download_poly_limbs": (C.c_int, [vp, vp, C.c_int, vpp]),
    "mkhe_ct_limbs": (C.c_int, [vp]),
    "mkhe_ct_nparties": (C.c_int, [vp]),
    "mkhe_ct_devptr": (vp, [vp]),
    "mkhe_buf_alloc": (C.c_int, [vp, C.c_size_t, vpp]),
    "mkhe_buf_free": (None, [vp, vp]),
    "mkhe_buf_upload": (C.c_int, [vp, vp, u64p, C.c_size_t]),
    "mkhe_buf_download": (C.c_int, [vp, vp, u64p, C.c_size_t]),
    "mkhe_ntt": (C.c_int, [vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
    "mkhe_decompose": (C.c_int, [vp, C.c_int, C.c_int, vp, C.c_int, vp]),
    "mkhe_hoisted_form": (C.c_int, [vp, C.c_int, vp, vpp]),
    "mkhe_external_product": (C.c_int, [vp, C.c_int, C.c_int, vp, C.c_int, vp, vp, C.c_int]),
    "mkhe_external_product_hoisted": (C.c_int, [vp, C.c_int, vp, vp, vp, C.c_int]),
    "mkhe_mul_and_relin": (C.c_int, [vp, vp, vp, vpp, vpp, vpp, vpp, vpp, vp, vp]),
    "mkhe_mul_relin_rescale": (C.c_int, [vp, vp, vp, vpp, vpp, vpp, vpp, vpp, vp, vp]),
    "mkhe_mr_partial": (C.c_int, [vp, vp, vp, vpp, vpp, vpp, vpp, C.c_int, vp, vp, vp]),
    "mkhe_swk_fold": (C.c_int, [vp, vp, C.c_int, C.c_int]),
    "mkhe_swk_fold_pieces": (C.c_int, [vp, vp, C.c_int, C.c_long, C.c_long, C.c_long, C.c_int, C.c_int, vp]),
    "mkhe_mr_finish": (C.c_int, [vp, vp, vp, vp, vp, vpp, vp, vp]),
    "mkhe_mr_finish_head": (C.c_int, [vp, vp, vp, vp, vp]),
    "mkhe_mr_finish_tail": (C.c_int, [vp, vp, vp, vp, vpp, vp, vp]),
    "mkhe_ct_fold": (C.c_int, [vp, vp]),
    "mkhe_rotate":(C.c_int, [vp, C.c_uint64, vp, vpp, vpp, vp, vp]),
    "mkhe_ctx_set_owned": (C.c_int, [vp, i32p, C.c_int]),
    "mkhe_lsh_phase": (C.c_int, [vp, C.c_int, vp, vp, vpp, vpp, vpp, vp, vp, vp, C.POINTER(C.c_size_t)]),
    "mkhe_rotate_partial": (C.c_int, [vp, vp, vpp, vpp, vp, C.c_int, vp]),
    "mkhe_ct_automorphism": (C.c_int, [vp, C.c_uint64, vp, vp]),
    "mkhe_conjugate": (C.c_int, [vp, C.c_uint64, vp, vpp, vp, vp]),
    "mkhe_rescale": (C.c_int, [vp, vp, C.c_int, vp]),
    "mkhe_ct_add": (C.c_int, [vp, vp, vp, vp]),
    "mkhe_ct_sub": (C.c_int, [vp, vp, vp, vp]),
    "mkhe_ct_mul_const": (C.c_int, [vp, vp, u64p, u64p, vp]),
    "mkhe_ct_mul_ptxt": (C.c_int, [vp, vp, vp, vp]),
    "mkhe_ct_create_batch": (C.c_int, [vp, C.c_int, C.c_int, i32p, C.c_int, vpp]),
    "mkhe_ct_destroy_batch": (None, [vp, C.c_int, vpp]),
    "mkhe_swk_create_batch": (C.c_int, [vp, C.c_int, vpp]),
    "mkhe_swk_destroy_batch": (None, [vp, C.c_int, vpp]),
    "mkhe_hoisted_form_batch": (C.c_int, [vp, C.c_int, C.c_int, vpp, vpp]),
    "mkhe_rotate_batch": (C.c_int, [vp, C.c_uint64, C.c_int, vpp, vpp, vpp, vp, vpp]),
    "mkhe_ct_sum": (C.c_int, [vp, C.c_int, vpp, vp]),
    "mkhe_rotate_multi": (C.c_int, [vp, C.c_int, u64p, vpp, vpp, vpp, vpp, vpp, vpp]),
    "mkhe_mul_relin_batch": (C.c_int, [vp, C.c_int, vpp, vpp, vpp, vpp, vpp, vpp, vpp, vp, C.c_int, vpp]),
    "mkhe_ct_binary_batch": (C.c_int, [vp, C.c_int, C.c_int, vpp, vpp, vpp]),
    "mkhe_ct_mul_ptxt_batch": (C.c_int, [vp, C.c_int, vpp, vp, C.c_int, vpp]),
    "mkhe_ctx_create_bfv": (C.c_int, [vpp, C.c_int, u64p, u64p, C.c_int, u64p, C.c_int, C.c_int, C.c_uint64, C.c_int]),
    "mkhe_bfv_modup_q_to_r": (C.c_int, [vp, vp, vp, C.c_int]),
    "mkhe_bfv_rescale": (C.c_int, [vp, vp, vp, C.c_int]),
    "mkhe_bfv_quantize": (C.c_int, [vp, vp, vp, C.c_int]),
    "mkhe_bfv_ntt_r": (C.c_int, [vp, vp, vp, C.c_int, C.c_int]),
    "mkhe_bfv_decompose": (C.c_int, [vp, vp, vp, vp]),
    "mkhe_bfv_external_product_hoisted": (C.c_int, [vp, vp, vp, vp, vp, vp]),
    "mkhe_bfv_external_product": (C.c_int, [vp, vp, vp, vp, vp]),
    "mkhe_bfv_mr_partial": (C.c_int, [vp, vp, vp, vpp, vpp, vpp, vpp, C.c_int, vp, vp, vp, vp, vp]),
    "mkhe_bfv_mr_finish": (C.c_int, [vp, vp, vp, vp, vp, vp, vp, vpp, vp, vp]),
    "mkhe_bfv_mul_relin": (C.c_int, [vp, vp, vp, vpp, vpp, vpp, vpp, vpp, vp, vp]),
    "mkhe_bfv_mul_relin_unhoisted": (C.c_int, [vp, vp, vp, vpp, vpp, vpp, vpp, vpp, vp, vp]),
    "mkhe_keygen_secret": (C.c_int, [vp, s32p, vp]),
    "mkhe_keygen_switching_key": (C.c_int, [vp, vp, s32p, vp]),
    "mkhe_keygen_public_key": (C.c_int, [vp, vp, s32p, vp, vp]),
    "mkhe_keygen_relin_key": (C.c_int, [vp, vp, vp, s32p, vp, vp, vp, vp, vp]),
    "mkhe_keygen_rotation_key": (C.c_int, [vp, C.c_uint64, vp, s32p, vp, vp]),
    "mkhe_keygen_conjugation_key": (C.c_int, [vp, vp, s32p, vp, vp]),
    "mkhe_bfv_keygen_switching_key": (C.c_int, [vp, vp, u64p, s32p, vp]),
    "mkhe_bfv_keygen_relin_key": (C.c_int, [vp, vp, vp, u64p, u64p, s32p, vp, vp, vp, vp, vp, vp, vp, vp]),
    "mkhe_crs_expand": (C.c_int, [vp, C.c_uint64, C.c_int32, vp]),
    "mkhe_prof_enable": (C.c_int, [vp, C.c_int]),
    "mkhe_ntt_trace": (C.c_int, [vp, vp]),
    "mkhe_set_overlap": (C.c_int, [vp, C.c_int]),
    "mkhe_ntt_choice": (C.c_int, [vp, C.c_long, C.c_int]),
    "mkhe_ctx_set_ntt_choice": (C.c_int, [vp, C.c_long, C.c_int, C.c_int]),
    "mkhe_ctx_set_batch_lanes": (C.c_int, [vp, C.c_long]),
    "mkhe_pool_held_bytes": (C.c_longlong, [vp]),
    "mkhe_pool_trim": (C.c_int, [vp]),
    "mkhe_f2_schedule_probe": (C.c_int, [C.c_int, C.c_int, C.c_int, C.POINTER(C.c_long), C.c_int, C.POINTER(C.c_ubyte), C.POINTER(C.c_int)]),
    "mkhe_prof_nclass": (C.c_int, []),
    "mkhe_prof_name": (C.c_char_p, [C.c_int]),
    "mkhe_prof_collect": (C.c_int, [vp, C.POINTER(C.c_double), C.POINTER(C.c_long), C.POINTER(C.c_double)]),
}


class MkheError(RuntimeError):
    """Raised where the reference panics / returns an error."""


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                "mkhe_kklss_amd: %s is missing -- build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950).  There is no CPU fallback." % LIB_PATH)
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            f = getattr(L, name)          # AttributeError if the library does not export it
            f.restype = res
            f.argtypes = args
        _lib = L
    return _lib


def check(rc):
    if rc != 0:
        raise MkheError(lib().mkhe_last_error().decode())


def handle_array(handles):
    """list of handles (or None) -> (void*)[]; None list -> NULL"""
    if handles is None:
        return None
    return (C.c_void_p * max(len(handles), 1))(*[h for h in handles])
