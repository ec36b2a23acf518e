"""CPU test of the launch harness itself: the UNMODIFIED reference train_syn.py runs under eld_amd.launch's shims
(SURVEY.md App. D).  Needs the reference checkout, so it is skipped on the GPU box where /root/reference is absent;
with --plugins none it exercises the harness, not the HIP path (which needs a GPU)."""
import os
import subprocess
import sys

import pytest

REF = '/root/reference'
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(not os.path.exists(os.path.join(REF, 'train_syn.py')), reason='reference checkout not present')
def test_unmodified_train_syn_imports_and_builds_pipeline(tmp_path):
    code = (
        "import sys, runpy; sys.path.insert(0, %r)\n"
        "import eld_amd.launch as L, os\n"
        "L.prepare_cwd(%r, %r); os.chdir(%r); sys.path.insert(0, %r); L.install_shims(patches=4, patch_hw=(64, 64))\n"
        "import noise, dataset.lmdb_dataset as ld, dataset.sid_dataset as sd, engine, models\n"
        "nm = noise.NoiseModel(model='Pg', include=4)\n"
        "clean = ld.LMDBDataset('data/Train/SID_Sony_Raw.db')\n"
        "ds = sd.ELDTrainDataset(target_dataset=clean, input_datasets=[sd.SynDataset(clean, noise_maker=nm)])\n"
        "d = ds[1]; assert d['input'].shape == (4, 64, 64) and d['input'].min() >= 0 and d['input'].max() <= 1\n"
        "assert 'eld_model' in models.__dict__ and 'unet' in models.arch.__dict__\n"
        "print('HARNESS_OK')\n" % (ROOT, REF, str(tmp_path), str(tmp_path), REF))
    r = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, timeout=300)
    assert 'HARNESS_OK' in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


def test_plugin_registries_are_importable():
    import eld_amd.launch as L
    import eld_amd.noise as n
    import eld_amd.unet as u
    import eld_amd.model as m
    assert callable(L.main) and callable(u.unet) and callable(m.eld_model) and hasattr(n, 'NoiseModel')
