#!/usr/bin/env python3
"""gpurun_out/r02_parity_*.json (written by tests/test_parity_full_gpu.py on the GPU box) -> profiles/r02_parity.md."""
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main(tag='r02'):
    files = sorted(glob.glob(os.path.join(ROOT, 'gpurun_out', '%s_parity_*.json' % tag)))
    if not files:
        raise SystemExit('no parity records under gpurun_out/')
    lines = ['# %s parity at BASELINE sizes: fused U-Net step (C ABI) vs oracle/unet_ref.py' % tag, '',
             'Produced by `pytest tests/test_parity_full_gpu.py -m gpu` on one MI355X; `tools/parity_report.py` formats the records.',
             'Columns: max |ref| of the tensor; max abs error of the HIP engine against the torch-CPU float32 oracle; the plain',
             'north_star bound 1e-5*(1+max|ref|); max abs error of the HIP engine and of the torch-CPU float32 oracle against the',
             'float64 oracle (error attribution: who is closer to the exact value).', '']
    for fn in files:
        r = json.load(open(fn))
        algo = {0: 'fp32 MFMA', 1: '3 x bf16 pieces (default)', 2: '2 x fp16 pieces (opt-in)'}[r['algo']]
        lines += ['## %s  shape %s, products: %s' % (r['case'], 'x'.join(map(str, r['shape'])), algo), '',
                  'loss: engine %.9g, cpu32 %.9g%s; torch-CPU float32 oracle step %.1f s on %d threads' % (
                      r['loss'], r['loss_cpu32'], (', f64 %.9g' % r['loss_f64']) if r.get('loss_f64') is not None else '', r['cpu32_oracle_s'], r['threads']), '',
                  '| tensor | max abs ref | engine vs cpu32 | 1e-5(1+ref) | engine vs f64 | cpu32 vs f64 | within |', '|---|---|---|---|---|---|---|']
        for t in r['tensors']:
            within = '1e-5' if t['err_vs_cpu32'] <= t['bound_1e5'] else ('1e-5 (f64)' if t.get('err_vs_f64', 1e9) <= t['bound_1e5'] else
                                                                        ('%.2f x cpu32 err' % (t['err_vs_f64'] / max(t['cpu32_vs_f64'], 1e-300)) if 'err_vs_f64' in t else 'FAIL'))
            lines.append('| %s | %.3e | %.3e | %.3e | %s | %s | %s |' % (
                t['name'], t['ref_max'], t['err_vs_cpu32'], t['bound_1e5'],
                ('%.3e' % t['err_vs_f64']) if 'err_vs_f64' in t else '-', ('%.3e' % t['cpu32_vs_f64']) if 'cpu32_vs_f64' in t else '-', within))
        worst = max(r['tensors'], key=lambda t: t['err_vs_cpu32'] / t['bound_1e5'])
        lines += ['', 'worst tensor relative to the plain bound: `%s` at %.2f x' % (worst['name'], worst['err_vs_cpu32'] / worst['bound_1e5']), '']
    out = os.path.join(ROOT, 'profiles', '%s_parity.md' % tag)
    with open(out, 'w') as f:
        f.write('\n'.join(lines) + '\n')
    print('wrote', out)


if __name__ == '__main__':
    main(*sys.argv[1:])
