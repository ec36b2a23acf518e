"""eld_amd -- MI355X (gfx950) native implementation of ELD's data-parallel hot path.

Per-pixel physics-based noise synthesis on packed-raw Bayer tensors feeding the SID U-Net
forward/backward, behind the reference's own plugin surface:

    eld_amd.noise.NoiseModel        <->  reference noise.NoiseModel          (noise.py:175-225)
    eld_amd.unet.unet / UNetSeeInDark <-> reference models.arch.unet          (models/arch/__init__.py:6-7)
    eld_amd.model.ELDModel / eld_model <-> reference models.eld_model          (models/__init__.py:3-4)
    eld_amd.engine.Engine            <->  reference engine.Engine             (engine.py:10-128)

All device work goes through the C ABI of eld_amd/libeld_amd.so (include/eld_amd.h), hand-written
HIP for gfx950.  There is no CPU fallback: if the library is missing, loading fails loudly.
"""
from ._lib import load_library, lib, LibraryMissing  # noqa: F401

__all__ = ['load_library', 'lib', 'LibraryMissing']
__version__ = '0.1.0'


FP32_PRODUCT_SCHEMES = {'mfma': 0, 'bf16x3': 1, 'fp16x2': 2}


def set_fp32_products(scheme):
    """How the fp32 convolutions form their products (process-wide; include/eld_amd.h eld_conv_fp32_algo): 'bf16x3' (default:
    exact three-piece bf16 split, 6 MFMA products), 'fp16x2' (two fp16 pieces behind per-tensor power-of-two scales, 22-bit
    products, 3 MFMA products, ~1.3x faster) or 'mfma' (v_mfma_f32_32x32x2_f32).  Returns the previous scheme's name."""
    prev = load_library().eld_conv_fp32_algo(FP32_PRODUCT_SCHEMES[scheme])
    return [k for k, v in FP32_PRODUCT_SCHEMES.items() if v == prev][0]
