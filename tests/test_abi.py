"""CPU tests of the drop-in boundary: the C-ABI library loads and exports every symbol the header
declares; the host-side plugin logic (no compute) behaves like the reference's."""
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_functions():
    src = open(os.path.join(ROOT, 'include', 'eld_amd.h')).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    return sorted(set(re.findall(r'\b(eld_[a-z0-9_]+)\s*\(', src)))


def test_library_exports_every_declared_symbol(eld_lib):
    from eld_amd import _lib
    names = header_functions()
    assert len(names) >= 7
    for n in names:
        assert hasattr(eld_lib, n), 'libeld_amd.so does not export %s' % n
        assert n in _lib.SIGNATURES, 'eld_amd/_lib.py does not bind %s' % n
    assert sorted(_lib.SIGNATURES) == names
    hdr = int(re.search(r'#define ELD_ABI_VERSION (\d+)', open(os.path.join(ROOT, 'include', 'eld_amd.h')).read()).group(1))
    assert eld_lib.eld_abi_version() == hdr == _lib.ABI_VERSION      # header, library and binding name the same ABI
    assert b'gfx950' in eld_lib.eld_build_info()
    assert b'EINVAL' in eld_lib.eld_error_string(-1)


def test_philox_round_count_agrees_everywhere(eld_lib):
    """The sampler's noise stream depends on the Philox round count: the library (csrc/philox.h ELD_PHILOX_ROUNDS), the package
    (eld_amd/_lib.py PHILOX_ROUNDS, checked at load) and the test oracle (oracle/philox_ref.py ROUNDS) must name the same number."""
    from eld_amd import _lib as L
    from oracle import philox_ref as px
    assert eld_lib.eld_philox_rounds() == L.PHILOX_ROUNDS == px.ROUNDS


def test_missing_library_fails_loudly(tmp_path):
    import eld_amd
    with pytest.raises(eld_amd.LibraryMissing):
        eld_amd.load_library(str(tmp_path / 'nope.so'))


def test_argument_errors_without_gpu(eld_lib):
    # argument validation happens before any device work, so these run on a CPU-only box
    assert eld_lib.eld_noise_forward(None, 0, None, None, -1, 4, 8, 8, 0, 0, None, None, None) == -1
    assert eld_lib.eld_noise_forward(None, 7, None, None, 1, 4, 8, 8, 0, 0, None, None, None) == -1
    assert eld_lib.eld_noise_forward(None, 0, None, None, 1, 4, 8, 8, 1 | 2, 0, None, None, None) == -1
    assert eld_lib.eld_noise_forward(None, 0, None, None, 1, 3, 8, 8, 16, 0, None, None, None) == -1   # row noise needs C==4
    assert eld_lib.eld_noise_forward(None, 0, None, None, 0, 4, 8, 8, 0, 0, None, None, None) == 0     # empty batch
    assert eld_lib.eld_noise_forward(None, 0, None, None, 1, 4, 0, 8, 0, 0, None, None, None) == 0     # empty image
    assert eld_lib.eld_noise_forward(None, 0, None, None, 1, 4, 8, 8, 0, 0, None, None, None) == -1    # null pointers


def test_params_record_layout():
    from eld_amd import _lib
    from eld_amd.noise import NoiseParams
    assert _lib.NOISE_PARAMS_DTYPE.itemsize == 64
    r = NoiseParams(2.5, 6.0, 15583, 200.0, tl_lambda=-0.1, tl_scale=3.0, row_scale=0.5, color_bias=(1, 2, 3, 4)).record((7 << 32) | 9)
    raw = np.frombuffer(r.tobytes(), dtype=np.float32)
    assert raw[:8].tolist() == [2.5, 6.0, np.float32(-0.1), 3.0, 0.5, 1.0, 15583.0, 200.0]
    assert raw[8:12].tolist() == [1, 2, 3, 4]
    assert np.frombuffer(r.tobytes(), dtype=np.uint32)[12:14].tolist() == [9, 7]
    K, g, s, ratio = NoiseParams(1.0, 2.0, 3, 4.0)            # unpacks like noise.py:225's tuple
    assert (K, g, s, ratio) == (1.0, 2.0, 3, 4.0)


def test_plugin_constructor_and_sample_params(golden_dir, capsys):
    """NoiseModel ctor semantics (noise.py:175-199) and _sample_params draw order (noise.py:201-225)."""
    from eld_amd.noise import NoiseModel, ALL_CAMERAS, model_flags
    recs = np.load(os.path.join(golden_dir, 'sample_params.npz'))['recs']
    i = 0
    for inc in (None, 4, 1):
        nm = NoiseModel(model='g', include=inc)
        assert nm.cameras == (ALL_CAMERAS if inc is None else [ALL_CAMERAS[inc]])
        for s in (0, 1, 2018):
            np.random.seed(s)
            for _ in range(3):
                got = nm._sample_params()
                assert np.array_equal(np.array(tuple(got), np.float64), recs[i][2:])
                i += 1
    out = capsys.readouterr().out
    assert '[i] NoiseModel with camera_params/release' in out and '[i] using noise model g' in out
    with pytest.raises(AssertionError):
        NoiseModel(include=1, exclude=2)
    with pytest.raises(AssertionError):
        NoiseModel(cfa='foveon')
    nm = NoiseModel(model='g', exclude=0)
    assert sorted(nm.cameras) == sorted(ALL_CAMERAS[1:])
    assert nm.model == 'g' and nm.raw_packer.cfa == 'bayer' and set(nm.camera_params) == set(nm.cameras)
    # letter parsing: 'P' shadows 'p' (noise.py:158-160)
    assert model_flags('Pg') == 1 | 4 and model_flags('P+g') == 1 | 4 and model_flags('pg') == 2 | 4
    assert model_flags('Ppg') == 1 | 4 and model_flags('PGRU') == 1 | 8 | 16 | 32 and model_flags('') == 0
    # full-model params: the reference's 4-tuple is unchanged for a given seed
    np.random.seed(5)
    base = NoiseModel(model='Pg', include=4)._sample_params()
    np.random.seed(5)
    full = NoiseModel(model='PGRU', include=4)._sample_params()
    assert tuple(base) == tuple(full) and full.tl_scale > 0 and full.row_scale > 0 and -0.25 < full.tl_lambda < 0.25


@pytest.mark.parametrize('src_name,kernel,wait', [('conv_bfs.hip', 'conv_bfs_kernelILi2ELi8ELi1EEEv8ConvArgs:', 's_waitcnt vmcnt(5)'),
                                                  ('conv_bfs.hip', 'conv_bfs_kernelILi2ELi8ELi2EEEv8ConvArgs:', 's_waitcnt vmcnt(5)'),
                                                  ('conv_bfw.hip', 'conv_bfw_kernelILb1EEEv8ConvArgs:', 's_waitcnt vmcnt(3)')])
def test_hand_issued_loads_are_not_touched_before_their_wait(src_name, kernel, wait):
    """conv_bfs_kernel<.., ACT = 1 / 2> and conv_bfw_kernel<ACT = true> load the saved activations (ACT = 2: their slope codes) with inline-asm global loads that hipcc
    does not count (cdna_hip_programming.md 5.7 item 1): between such a load and the hand-written `s_waitcnt vmcnt(n)` that retires it the
    compiler must neither read nor copy nor overwrite the destination registers.  Audit of the generated gfx950 assembly (cross-compiles
    without a GPU); silent corruption otherwise -- a passing numerical test is not evidence for this hazard."""
    import re
    import shutil
    import subprocess
    import tempfile
    hipcc = shutil.which('hipcc') or '/opt/rocm/bin/hipcc'
    if not os.path.exists(hipcc):
        pytest.skip('no hipcc')
    src = os.path.join(ROOT, 'eld_amd', 'csrc', src_name)
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, 'k.s')
        subprocess.check_call([hipcc, '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-ffp-contract=off', '-fhip-fp32-correctly-rounded-divide-sqrt',
                               '-S', '--cuda-device-only', '-o', out, src], stderr=subprocess.DEVNULL)
        text = open(out).read()
    body = text[text.index(kernel):]
    body = body[:body.index('.Lfunc_end')]
    # basic blocks (labels, branches) and a forward dataflow of the "loaded, not yet waited for" register set: mutually exclusive branches
    # may reuse each other's destination registers as temporaries, so a linear scan of the text would report false hazards
    blocks, cur, in_asm = [], {'label': None, 'ins': []}, False
    for ln in body.split('\n')[1:]:
        t = ln.strip()
        if t.startswith(';;#ASMSTART'):
            in_asm = True
            continue
        if t.startswith(';;#ASMEND'):
            in_asm = False
            continue
        m = re.match(r'^(\.LBB\d+_\d+):', t)
        if m:
            blocks.append(cur)
            cur = {'label': m.group(1), 'ins': []}
            continue
        if not t or t.startswith(';') or t.startswith('.'):
            continue
        code = t.split(';')[0].strip()
        cur['ins'].append((in_asm, code))
        if not in_asm and (code.startswith('s_branch') or code.startswith('s_cbranch') or code.startswith('s_endpgm')):
            blocks.append(cur)
            cur = {'label': None, 'ins': []}
    blocks.append(cur)
    blocks = [b for b in blocks if b['ins'] or b['label']]
    index = {b['label']: i for i, b in enumerate(blocks) if b['label']}
    succ = []
    for i, b in enumerate(blocks):
        last = b['ins'][-1][1] if b['ins'] else ''
        if last.startswith('s_endpgm'):
            succ.append([])
        elif last.startswith('s_branch'):
            succ.append([index[last.split()[1]]])
        elif last.startswith('s_cbranch'):
            succ.append([index[last.split()[1]]] + ([i + 1] if i + 1 < len(blocks) else []))
        else:
            succ.append([i + 1] if i + 1 < len(blocks) else [])

    def regs_of(code):
        regs = set(int(r) for r in re.findall(r'\bv(\d+)\b', code))
        for a_, b_ in re.findall(r'v\[(\d+):(\d+)\]', code):
            regs |= set(range(int(a_), int(b_) + 1))
        return regs

    def undef_high_addend(code, pending):
        """hipcc's 32-bit multiply-add on gfx9 is v_mad_u64_u32 / v_mad_i64_i32 with a 64-bit addend pair whose HIGH register is an undefined
        placeholder (only the low half of the result is used) -- any register will do for it, also one that holds an in-flight load's destination:
        a read whose value cannot reach the used half of the result.  Nothing else is excused: the pending register must be exactly the addend's
        high half and appear nowhere else in the instruction (in particular not in the destination pair)."""
        m = re.match(r'v_mad_[ui]64_[ui]32 v\[(\d+):(\d+)\], (?:s\[\d+:\d+\]|vcc), (\S+), (\S+), v\[(\d+):(\d+)\]$', code)
        if not m:
            return False
        hi = int(m.group(6))
        others = set(range(int(m.group(1)), int(m.group(2)) + 1)) | regs_of(m.group(3)) | regs_of(m.group(4)) | {int(m.group(5))}
        return (regs_of(code) & pending) == {hi} and hi not in others

    def reads_writes(code):
        """(registers read, registers written) of one compiler instruction: the first operand is the destination except for stores / compares
        into SGPRs (which write no VGPR); every other VGPR mentioned is a source.  Conservative: a destination that is also a source counts as read."""
        ops = code.split(None, 1)
        if len(ops) < 2:
            return set(), set()
        parts = [p_.strip() for p_ in ops[1].split(',')]
        mnem = ops[0]
        no_vdst = mnem.startswith(('global_store', 'buffer_store', 'ds_write', 'ds_store', 's_', 'v_cmp', 'v_cmpx', 'v_readfirstlane', 'v_readlane', 'scratch_store'))
        wr = set() if no_vdst else regs_of(parts[0])
        rd = set()
        for p_ in (parts if no_vdst else parts[1:]):
            rd |= regs_of(p_)
        return rd, wr

    def high_half_dead(block_idx, pos, dst_hi):
        """The exemption above holds only if the HIGH half of the multiply-add's 64-bit result is never used: from the instruction on, every path
        must overwrite v<dst_hi> before reading it (or end)."""
        seen, work = set(), [(block_idx, pos + 1)]
        while work:
            bi, start = work.pop()
            if (bi, start) in seen:
                continue
            seen.add((bi, start))
            ins = blocks[bi]['ins']
            killed = False
            for is_asm, code in ins[start:]:
                rd, wr = (regs_of(code), set()) if is_asm else reads_writes(code)
                if dst_hi in rd:
                    return False
                if dst_hi in wr:
                    killed = True
                    break
            if not killed:
                work.extend((j, 0) for j in succ[bi])
        return True

    pend_in = [set() for _ in blocks]
    checked, violations = 0, []
    changed = True
    while changed:
        changed = False
        checked, violations = 0, []
        for i, b in enumerate(blocks):
            pending = set(pend_in[i])
            for pos, (is_asm, code) in enumerate(b['ins']):
                if is_asm:
                    m = re.match(r'global_load_dwordx4 v\[(\d+):(\d+)\]', code)
                    m1 = re.match(r'global_load_dword v(\d+),', code)
                    if m:
                        pending |= set(range(int(m.group(1)), int(m.group(2)) + 1))
                    elif m1:
                        pending.add(int(m1.group(1)))
                    elif code.startswith(wait):
                        if pending:
                            checked += 1
                        pending = set()
                    continue
                if pending and (regs_of(code) & pending):
                    ok = undef_high_addend(code, pending)
                    if ok:      # ... and the result's high half (the only place the placeholder's value can reach) must be dead
                        mm = re.match(r'v_mad_[ui]64_[ui]32 v\[(\d+):(\d+)\]', code)
                        ok = high_half_dead(i, pos, int(mm.group(2)))
                    if not ok:
                        violations.append(code)
            for j in succ[i]:
                if not pending <= pend_in[j]:
                    pend_in[j] |= pending
                    changed = True
    assert not violations, 'compiler instructions touch un-waited asm load destinations: %s' % violations[:5]
    assert checked >= 1


def test_traffic_constant_is_tied_to_the_library_sources(eld_lib, tmp_path, monkeypatch):
    """bench.py reports the committed PMC byte counts (profiles/traffic.json) only for a library built from the sources those
    passes ran on: eld_build_info() carries the source hash (__graft_entry__.source_hash), traffic.json records it, a mismatch drops the figure."""
    import importlib.util
    import json
    import __graft_entry__ as ge
    from eld_amd import _lib as L
    have = L.build_src_hash()
    assert have == ge.source_hash() and len(have) == 16, (have, ge.source_hash())      # the in-tree library is a build of the in-tree sources
    spec = importlib.util.spec_from_file_location('bench_mod', os.path.join(ROOT, 'bench.py'))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    (tmp_path / 'profiles').mkdir()
    monkeypatch.setattr(bench, 'ROOT', str(tmp_path))
    base = {'unet_conv_bytes_per_pass': 1.0, 'unet_conv_bytes_per_pass_bf16': 2.0, 'sampler_bytes_per_pixel': 8.5, 'frames_per_pass': 8}
    json.dump(dict(base, library_src_hash=have, library_src_hash_bf16='0' * 16), open(tmp_path / 'profiles' / 'traffic.json', 'w'))
    t = bench.load_traffic()
    assert t['unet_conv_bytes_per_pass'] == 1.0 and 'sampler_bytes_per_pixel' in t
    assert 'unet_conv_bytes_per_pass_bf16' not in t and 'not reported' in t['_stale']['unet_conv_bytes_per_pass_bf16']
    json.dump(base, open(tmp_path / 'profiles' / 'traffic.json', 'w'))                  # a file from before the hash existed: nothing is reported
    t = bench.load_traffic()
    assert 'unet_conv_bytes_per_pass' not in t and 'sampler_bytes_per_pixel' not in t and len(t['_stale']) == 2
