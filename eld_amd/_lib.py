"""ctypes binding of include/eld_amd.h (the drop-in C ABI).  No torch types cross the ABI:
device pointers travel as integers, the stream as the raw hipStream_t."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('ELD_AMD_LIB') or os.path.join(_HERE, 'libeld_amd.so')      # ELD_AMD_LIB: developer builds (tools/build_dev.sh)

# flags / enums of include/eld_amd.h
SHOT_POISSON, SHOT_GAUSS, READ_GAUSS, READ_TL, ROW, QUANT, CBIAS, CLIP = 1, 2, 4, 8, 16, 32, 64, 128
AUG_NOTRANSPOSE = 256
IN_F32, IN_U16 = 0, 1
NPLANES = 6
PLANE = {'counts': 0, 'n_shot': 1, 'n_read': 2, 't_tl': 3, 'n_row': 4, 'u_q': 5}

# numpy mirror of struct EldNoiseParams (64 bytes)
NOISE_PARAMS_DTYPE = np.dtype([
    ('K', '<f4'), ('g_scale', '<f4'), ('tl_lambda', '<f4'), ('tl_scale', '<f4'), ('row_scale', '<f4'),
    ('q_step', '<f4'), ('saturation', '<f4'), ('ratio', '<f4'), ('color_bias', '<f4', (4,)),
    ('sample_id_lo', '<u4'), ('sample_id_hi', '<u4'), ('reserved', '<u4', (2,))])
assert NOISE_PARAMS_DTYPE.itemsize == 64

_vp, _i, _u32, _u64, _f, _d, _sz = C.c_void_p, C.c_int, C.c_uint32, C.c_uint64, C.c_float, C.c_double, C.c_size_t

# name -> (restype, argtypes): exactly the prototypes of include/eld_amd.h
SIGNATURES = {
    'eld_abi_version': (_i, []),
    'eld_build_info': (C.c_char_p, []),
    'eld_error_string': (C.c_char_p, [_i]),
    'eld_noise_forward': (_i, [_vp, _i, _vp, _vp, _i, _i, _i, _i, _u32, _u64, _vp, _vp, _vp]),
    'eld_noise_forward_strided': (_i, [_vp, _i, _sz, _vp, _sz, _vp, _i, _i, _i, _i, _u32, _u64, _vp, _vp, _vp]),
    'eld_augment_u16': (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _u32, _vp]),
    'eld_philox_rounds': (_i, []),
    'eld_philox_words': (_i, [_vp, _u32, _u32, _u64, _u32, _u32, _u64, _vp]),
    'eld_pack_bayer': (_i, [_vp, _vp, _i, _i, _i, _vp]),
    'eld_unpack_bayer': (_i, [_vp, _vp, _i, _i, _i, _vp]),
    'eld_pack_xtrans': (_i, [_vp, _vp, _i, _i, _i, _vp]),
    'eld_unpack_xtrans': (_i, [_vp, _vp, _i, _i, _i, _vp]),
    'eld_pack_raw_bayer_u16': (_i, [_vp, _vp, _i, _i, _i, C.POINTER(C.c_int), C.POINTER(C.c_float), _f, _vp]),
    'eld_augment': (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _u32, _vp]),
    'eld_unet_param_offsets': (_i, [_i, _i, _vp]),
    'eld_unet_workspace_bytes': (_sz, [_i, _i, _i, _i, _i]),
    'eld_unet_forward': (_i, [_vp, _vp, _vp, _vp, _sz, _i, _i, _i, _i, _i, _vp]),
    'eld_unet_forward_bf16': (_i, [_vp, _vp, _vp, _vp, _sz, _i, _i, _i, _i, _i, _vp]),
    'eld_unet_backward': (_i, [_vp, _vp, _vp, _vp, _sz, _i, _i, _i, _i, _i, _vp]),
    'eld_unet_backward_bf16': (_i, [_vp, _vp, _vp, _vp, _sz, _i, _i, _i, _i, _i, _vp]),
    'eld_unet_backward_buckets': (_i, [_vp, _vp, _vp, _vp, _sz, _i, _i, _i, _i, _i, _i, _vp, _vp, _i, _vp]),
    'eld_unet_forward_ex': (_i, [_vp, _vp, _vp, _vp, _sz, _i, _i, _i, _i, _i, _i, _i, _vp]),
    'eld_unet_infer_ex': (_i, [_vp, _vp, _vp, _vp, _sz, _i, _i, _i, _i, _i, _i, _i, _vp]),
    'eld_unet_backward_ex': (_i, [_vp, _vp, _vp, _vp, _sz, _i, _i, _i, _i, _i, _i, _i, _vp, _vp, _i, _vp]),
    'eld_unet_forward_loss_ex': (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _sz, _i, _i, _i, _i, _i, _i, _i, _i, _f, _vp]),
    'eld_conv_fp32_algo': (_i, [_i]),
    'eld_debug_conv_prof': (None, [_vp]),
    'eld_debug_kernel_mask': (_i, [_i]),
    'eld_debug_ws_state_entries': (_i, []),
    'eld_isp_process': (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _f, _vp, _vp, _i, _vp]),
    'eld_quality_assess_workspace_bytes': (_sz, [_i, _i, _i, _i]),
    'eld_quality_assess': (_i, [_vp, _vp, _vp, _vp, _sz, _i, _i, _i, _i, _f, _vp]),
    'eld_quality_assess_images': (_i, [_vp, _vp, _vp, _vp, _sz, _i, _i, _i, _i, _f, _vp]),
    'eld_illuminance_correct_workspace_bytes': (_sz, [_i]),
    'eld_illuminance_correct': (_i, [_vp, _vp, _vp, _vp, _sz, _i, _i, _sz, _vp]),
    'eld_l1_workspace_bytes': (_sz, []),
    'eld_l1_loss': (_i, [_vp, _vp, _vp, _vp, _vp, _sz, _f, _vp]),
    'eld_mse_loss': (_i, [_vp, _vp, _vp, _vp, _vp, _sz, _f, _vp]),
    'eld_adam_step': (_i, [_vp, _vp, _vp, _vp, _sz, _d, _d, _d, _d, _d, _i, _d, _vp]),
    'eld_layer_workspace_bytes': (_sz, [_i, _i, _i, _i, _i]),
    'eld_conv3x3_forward': (_i, [_vp, _i, _vp, _i, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _sz, _vp]),
    'eld_conv3x3_backward_data': (_i, [_vp, _vp, _vp, _vp, _i, _vp, _vp, _i, _i, _i, _i, _i, _vp, _sz, _vp]),
    'eld_conv3x3_backward_weight': (_i, [_vp, _vp, _i, _vp, _i, _vp, _vp, _i, _i, _i, _i, _vp, _sz, _vp]),
    'eld_convt2x2_forward': (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _sz, _vp]),
    'eld_convt2x2_backward_data': (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _sz, _vp]),
    'eld_convt2x2_backward_weight': (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _sz, _vp]),
    'eld_maxpool2x2_forward': (_i, [_vp, _vp, _i, _i, _i, _i, _vp]),
    'eld_maxpool2x2_backward': (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
}


ABI_VERSION = 2         # ELD_ABI_VERSION of include/eld_amd.h this binding was written against
PHILOX_ROUNDS = 7      # the sampler's generator: Philox4x32-7 (csrc/philox.h); checked against the library at load


class LibraryMissing(RuntimeError):
    pass


_lib = None


def load_library(path=None):
    """Load libeld_amd.so and bind every prototype.  Fails loudly -- there is no fallback path."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or LIB_PATH
    if not os.path.exists(p):
        raise LibraryMissing(
            '%s not found: the HIP extension has not been built.  Run `python __graft_entry__.py` '
            '(hipcc --offload-arch=gfx950) first; eld_amd has no CPU fallback.' % p)
    # PyTorch-ROCm ships its own copy of the HIP runtime; device memory, streams and events are torch's, so libeld_amd must
    # bind to THAT runtime instance: import torch first (a libamdhip64 loaded by us before torch's is a second runtime with
    # no device context -- every launch then fails with hipErrorNoDevice).
    import torch  # noqa: F401
    lib_ = C.CDLL(p)
    any_philox = bool(os.environ.get('ELD_AMD_ANY_PHILOX'))
    lib_.eld_abi_version.restype = C.c_int
    if lib_.eld_abi_version() != ABI_VERSION and not any_philox:      # before binding: an older library fails here, not with a missing-symbol AttributeError
        raise RuntimeError('libeld_amd ABI version %d, this package binds version %d of include/eld_amd.h: rebuild with `python __graft_entry__.py`'
                           % (lib_.eld_abi_version(), ABI_VERSION))
    for name, (res, args) in SIGNATURES.items():
        if name == 'eld_philox_rounds' and any_philox and not hasattr(lib_, name):
            continue                      # dev A/B runs against a library older than the symbol (tools/build_variant.sh <old rev>)
        fn = getattr(lib_, name)          # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    if not any_philox and lib_.eld_philox_rounds() != PHILOX_ROUNDS:
        # the noise stream of a (seed, sample id) pair depends on the round count: a library built with another -DELD_PHILOX_ROUNDS replays
        # different noise for the same checkpoint / seed ("noise streams")
        raise RuntimeError('libeld_amd runs Philox4x32-%d, this package expects %d rounds (set ELD_AMD_ANY_PHILOX=1 to accept)'
                           % (lib_.eld_philox_rounds(), PHILOX_ROUNDS))
    _lib = lib_
    return lib_


def lib():
    return load_library()


def build_src_hash():
    """The source hash compiled into the loaded library (eld_build_info: "... src=<16 hex digits>"; "unknown" for builds that did not pass one)."""
    info = lib().eld_build_info().decode()
    return info.rsplit('src=', 1)[1].strip() if 'src=' in info else 'unknown'


class EldError(RuntimeError):
    pass


def check(rc, what=''):
    if rc != 0:
        msg = lib().eld_error_string(int(rc)).decode()
        raise EldError('%s failed: %s (code %d)' % (what or 'libeld_amd call', msg, rc))


def dptr(t):
    """Device pointer of a torch tensor (or None)."""
    return None if t is None else C.c_void_p(t.data_ptr())


def cur_stream():
    import torch
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)
