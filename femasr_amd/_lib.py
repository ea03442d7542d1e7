"""ctypes binding of libfemasr_hip.so (C ABI declared in include/femasr_hip.h).

There is NO fallback: if the shared library is missing or a call fails, this
module raises — the product path never silently degrades to a CPU / eager path.
Build the library with `python femasr_amd/csrc/build.py` (hipcc, gfx950).
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
SO_PATH = os.path.join(_HERE, 'csrc', 'libfemasr_hip.so')

c_int, c_i64, c_f32 = ctypes.c_int, ctypes.c_int64, ctypes.c_float
vp, szt = ctypes.c_void_p, ctypes.c_size_t


class FemasrError(RuntimeError):
    pass


MAX_CODEBOOKS = 3


class Config(ctypes.Structure):
    _fields_ = ([(n, ctypes.c_int32) for n in ('in_channel', 'gt_resolution', 'lq_stage', 'scale_factor', 'use_quantize',
                                               'use_residual', 'n_codebooks')] +
                [(n, ctypes.c_int32 * MAX_CODEBOOKS) for n in ('codebook_scale', 'n_e', 'e_dim')] +
                [('device', ctypes.c_int32)])


class ConvArgs(ctypes.Structure):
    _fields_ = [
        ('in_', vp), ('B', ctypes.c_int32), ('H', ctypes.c_int32), ('W', ctypes.c_int32), ('Cin', ctypes.c_int32),
        ('w', vp), ('bias', vp),
        ('Cout', ctypes.c_int32), ('ksz', ctypes.c_int32), ('stride', ctypes.c_int32), ('pad', ctypes.c_int32),
        ('up2', ctypes.c_int32), ('prologue', ctypes.c_int32),
        ('pro_a', vp), ('pro_b', vp), ('pro_c', vp),
        ('act', ctypes.c_int32),
        ('res1', vp), ('res2', vp), ('out', vp),
        ('Ho', ctypes.c_int32), ('Wo', ctypes.c_int32),
        ('w_bf16x3', vp), ('gn_part', vp), ('w_up2', vp), ('w_wino', vp), ('fast_act', ctypes.c_int32), ('in_add', vp), ('w_bf16s', vp),
    ]


ABI_VERSION = 102      # femasr_version(): debug hooks in their own header, uint8 tile kernels (femasr_conv_args ends with w_bf16s)
PRO_NONE, PRO_GN_SILU, PRO_LN = 0, 1, 2
ACT_NONE, ACT_GELU = 0, 1

# name -> (restype, argtypes); every symbol include/femasr_hip.h declares
SIGNATURES = {
    'femasr_last_error': (ctypes.c_char_p, []),
    'femasr_version': (c_int, []),
    'femasr_create': (c_int, [ctypes.POINTER(Config), ctypes.POINTER(vp)]),
    'femasr_destroy': (None, [vp]),
    'femasr_num_weights': (c_int, [vp]),
    'femasr_weight_info': (c_int, [vp, c_int, ctypes.POINTER(ctypes.c_char_p), ctypes.POINTER(c_i64 * 4), ctypes.POINTER(c_int)]),
    'femasr_set_weight': (c_int, [vp, ctypes.c_char_p, vp, ctypes.POINTER(c_i64), c_int]),
    'femasr_finalize_weights': (c_int, [vp]),
    'femasr_set_streams': (c_int, [vp, c_int]),
    'femasr_workspace_bytes': (c_int, [vp, c_int, c_int, c_int, c_int, ctypes.POINTER(szt)]),
    'femasr_forward_shapes': (c_int, [vp, c_int, c_int, c_int, ctypes.POINTER(c_int), ctypes.POINTER(c_int), ctypes.POINTER(c_int),
                                      ctypes.POINTER(c_int * MAX_CODEBOOKS), ctypes.POINTER(c_int * MAX_CODEBOOKS)]),
    'femasr_forward': (c_int, [vp, vp, vp, c_int, c_int, c_int, c_int, vp, vp, vp, szt]),
    'femasr_forward_u8': (c_int, [vp, vp, vp, c_int, c_int, c_int, c_int, c_int, vp, vp, vp, szt]),
    'femasr_pad_u8hwc_to_nhwc': (c_int, [vp, vp, c_int, c_int, c_int, c_int, c_int, c_int, vp]),
    'femasr_crop_nhwc_to_u8hwc': (c_int, [vp, vp, c_int, c_int, c_int, c_int, c_int, c_int, vp]),
    'femasr_decode_workspace_bytes': (c_int, [vp, c_int, c_int, c_int, ctypes.POINTER(szt)]),
    'femasr_decode_indices': (c_int, [vp, vp, vp, c_int, c_int, c_int, vp, vp, szt]),
    'femasr_profile_enable': (c_int, [vp, c_int]),
    'femasr_profile_reset': (c_int, [vp]),
    'femasr_profile_slots': (c_int, [vp]),
    'femasr_profile_name': (ctypes.c_char_p, [vp, c_int]),
    'femasr_profile_get': (c_int, [vp, c_int, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(c_i64),
                                   ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double)]),
    'femasr_pad_nchw_to_nhwc': (c_int, [vp, vp, c_int, c_int, c_int, c_int, c_int, c_int, vp]),
    'femasr_crop_nhwc_to_nchw': (c_int, [vp, vp, c_int, c_int, c_int, c_int, c_int, c_int, vp]),
    'femasr_conv2d': (c_int, [vp, ctypes.POINTER(ConvArgs)]),
    'femasr_gn_coeffs': (c_int, [vp, vp, c_int, c_int, c_int, c_int, c_int, vp, vp, c_f32, vp, vp, vp]),
    'femasr_gn_coeffs_from_partials': (c_int, [vp, vp, c_int, c_int, c_int, c_int, c_int, c_int, vp, vp, c_f32, vp, vp]),
    'femasr_ln_stats': (c_int, [vp, vp, c_i64, c_int, c_f32, vp]),
    'femasr_layernorm': (c_int, [vp, vp, c_i64, c_int, vp, vp, c_f32, vp]),
    'femasr_gn_scratch_bytes': (szt, [c_int, c_int, c_int, c_int, c_int]),
    'femasr_window_attention': (c_int, [vp, vp, c_int, c_int, c_int, c_int, c_int, c_int, vp, vp]),
    'femasr_vq': (c_int, [vp, vp, c_i64, c_int, vp, vp, vp, c_int, vp, vp, vp]),
    'femasr_row_sqsum': (c_int, [vp, vp, c_i64, c_int, vp]),
    'femasr_vq_twopass_ok': (c_int, [c_int, c_int]),
    'femasr_vq_aux_bytes': (szt, [c_int, c_int]),
    'femasr_vq_prepare': (c_int, [vp, vp, vp, c_int, c_int, vp]),
    'femasr_vq_scratch_bytes': (szt, [c_i64, c_int]),
    'femasr_vq_twopass': (c_int, [vp, vp, c_i64, c_int, vp, vp, vp, c_int, vp, vp, vp]),
    'femasr_vq_candidates': (c_int, [vp, vp, c_i64, c_int, vp, vp, c_int, vp, vp]),
    'femasr_codebook_gather': (c_int, [vp, vp, c_i64, c_int, vp, c_int, vp]),
    'femasr_extract_tiles': (c_int, [vp, vp, c_int, c_int, c_int, c_int, vp, c_int, c_int, c_int, vp]),
    'femasr_paste_tiles': (c_int, [vp, vp, c_int, c_int, c_int, c_int, c_int, vp, c_int, c_int, c_int, vp]),
    'femasr_extract_tiles_u8': (c_int, [vp, vp, c_int, c_int, c_int, vp, c_int, c_int, c_int, vp]),
    'femasr_paste_tiles_u8': (c_int, [vp, vp, c_int, c_int, c_int, c_int, vp, c_int, c_int, c_int, vp]),
    'femasr_concat_resize': (c_int, [vp, vp, c_int, vp, c_int, c_int, c_int, c_int, c_int, c_int, vp]),
    'femasr_repack_oihw': (c_int, [vp, vp, c_int, c_int, c_int, c_int, vp]),
    'femasr_packed_weight_floats': (szt, [c_int, c_int, c_int, c_int]),
    'femasr_up2_weight_floats': (szt, [c_int, c_int]),
    'femasr_wino_weight_floats': (szt, [c_int, c_int]),
    'femasr_repack_oihw_wino': (c_int, [vp, vp, c_int, c_int, vp]),
    'femasr_wino_up2_weight_floats': (szt, [c_int, c_int]),
    'femasr_repack_oihw_wino_up2': (c_int, [vp, vp, c_int, c_int, vp]),
    'femasr_repack_oihw_up2': (c_int, [vp, vp, c_int, c_int, vp]),
    'femasr_packed_weight_bf16x3_bytes': (szt, [c_int, c_int, c_int, c_int]),
    'femasr_repack_oihw_bf16x3': (c_int, [vp, vp, c_int, c_int, c_int, c_int, vp]),
    'femasr_set_decoder_math': (c_int, [vp, c_int]),
    'femasr_set_linear_math': (c_int, [vp, c_int]),
    'femasr_packed_weight_bf16s_bytes': (szt, [c_int, c_int]),
    'femasr_repack_k1_bf16s': (c_int, [vp, vp, c_int, c_int, vp]),
    'femasr_packed_weight_conv3x3_bf16s_bytes': (szt, [c_int, c_int]),
    'femasr_repack_oihw_bf16s': (c_int, [vp, vp, c_int, c_int, vp]),
    'femasr_gn_silu_apply': (c_int, [vp, vp, c_int, c_int, c_int, c_int, vp, vp, vp]),
    'femasr_image_u8_to_f32': (c_int, [vp, vp, c_int, c_int, c_int, vp]),
    'femasr_image_f32_to_u8': (c_int, [vp, vp, c_int, c_int, c_int, vp]),
    'femasr_clock_probe': (c_int, [vp, c_int, vp]),
    'femasr_clock_probe_entries': (c_int, []),
}
# test / measurement hooks: include/femasr_hip_debug.h (exported by the library, not part of the drop-in interface)
DEBUG_SIGNATURES = {
    'femasr_debug_set_wino_limits': (c_int, [vp, c_int, c_int]),
    'femasr_debug_mfma_bf16': (c_int, [vp, vp, vp, vp, c_int, vp]),
    'femasr_gemm_force_config': (c_int, [c_int]),
    'femasr_conv_small_launch_blocks': (c_int, [c_int]),
}

_lib = None


def load():
    """Load the shared library and bind every declared symbol; raises if anything is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(SO_PATH):
        raise FemasrError(
            f'{SO_PATH} not found: the HIP extension is not built. Run `python femasr_amd/csrc/build.py` '
            '(needs hipcc; cross-compiles gfx950 without a GPU). There is no CPU fallback.')
    lib = ctypes.CDLL(SO_PATH)
    lib.femasr_version.restype = c_int
    if lib.femasr_version() != ABI_VERSION:
        raise FemasrError(f'{SO_PATH} reports ABI version {lib.femasr_version()}, this binding is written for {ABI_VERSION} '
                          '(femasr_conv_args layout): rebuild with `python femasr_amd/csrc/build.py --force`')
    for name, (res, args) in list(SIGNATURES.items()) + list(DEBUG_SIGNATURES.items()):
        fn = getattr(lib, name)      # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc):
    if rc != 0:
        msg = load().femasr_last_error()
        raise FemasrError(f'libfemasr_hip error {rc}: {msg.decode() if msg else "?"}')


def ptr(t):
    """Device pointer of a torch tensor (or None)."""
    return None if t is None else ctypes.c_void_p(t.data_ptr())
