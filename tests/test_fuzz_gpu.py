"""GPU: random (N, H, W) through the whole U-Net step -- the strip tiling, the free tile shapes and the DMA-staged weight gradient against the kernel
families they did not touch (tools/fuzz_shapes.py: bf16 specialised kernels == generic conv_igemm_kernel<bf16> bit for bit, wgrad8d ~ wgrad8, the
default fp32 scheme against the float64 oracle."""
import importlib.util
import os

import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip('torch')
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize('seed,big', [(3, 1), (11, 3)])
def test_random_shapes_agree_across_kernel_families(eld_lib, seed, big):
    spec = importlib.util.spec_from_file_location('fuzz_shapes', os.path.join(ROOT, 'tools', 'fuzz_shapes.py'))
    fz = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fz)
    assert fz.main(n_cases=8, seed=seed, big=big) == 0
