#!/usr/bin/env python3
"""gpurun_out/<tag>_parity_*.json (written by tests/test_parity_full_gpu.py on the GPU box) -> profiles/<tag>_parity.md."""
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main(tag='r06'):
    files = sorted(glob.glob(os.path.join(ROOT, 'gpurun_out', '%s_parity_*.json' % tag)))
    if not files:
        raise SystemExit('no parity records under gpurun_out/')
    lines = ['# %s parity at BASELINE sizes: fused U-Net step (C ABI) vs oracle/unet_ref.py' % tag, '',
             'Produced by `pytest tests/test_parity_full_gpu.py -m gpu` on one MI355X; `tools/parity_report.py` formats the records.',
             'Columns: max |ref| of the tensor; max abs error of the HIP engine against the torch-CPU float32 oracle; the plain',
             'north_star bound 1e-5*(1+max|ref|); max abs error of the HIP engine and of the torch-CPU float32 oracle against the',
             'float64 oracle (error attribution: who is closer to the exact value); `rel` = max abs error against the float64 oracle',
             '(cpu32 where there is none) divided by max |ref| -- the RELATIVE criterion (<= 2e-4, or within 8x of cpu32-vs-f64) every',
             'tensor must meet besides the absolute bound, so that the deep layers (max |ref| ~ 1e-10) are actually checked.', '']
    for fn in files:
        r = json.load(open(fn))
        if r['case'] == 'bench_batch8':
            lines += ['## bench_batch8  shape %s (the batch bench.py runs), products: default scheme' % 'x'.join(map(str, r['shape'])), '',
                      'gradients of the 8-frame launch chain against the MEAN of the eight single-frame gradient buffers (each single frame is the '
                      'oracle-pinned `frame1424x2128` case); outputs equal the single-frame outputs bit for bit.  loss (batch) %.9g, mean of single-frame '
                      'losses %.9g' % (r['loss8'], r['mean_single_loss']), '',
                      '| tensor | max abs ref | batch vs mean of single frames | 1e-5(1+ref) | rel | ok |', '|---|---|---|---|---|---|']
            for t in r['tensors']:
                lines.append('| %s | %.3e | %.3e | %.3e | %.1e | %s |' % (t['name'], t['ref_max'], t['err_vs_mean_of_single_frames'], t['bound_1e5'],
                                                                    t.get('rel_err', float('nan')), 'yes' if t['ok'] else 'NO'))
            lines.append('')
            continue
        if r['case'].startswith('bf16_'):
            lines += ['## %s  shape %s, BASELINE configs[2] bf16 engine vs the FLOAT64 oracle' % (r['case'], 'x'.join(map(str, r['shape']))), '',
                      'loss: engine %.9g, f64 oracle %.9g (rel %.2e); output PSNR vs f64 oracle %.2f dB' % (
                          r['loss'], r['loss_f64'], abs(r['loss'] - r['loss_f64']) / abs(r['loss_f64']), r['output_psnr_db']), '',
                      '| tensor | max abs ref | cosine | relative L2 error | max abs error |', '|---|---|---|---|---|']
            for t in r['tensors']:
                lines.append('| %s | %.3e | %.6f | %.3e | %.3e |' % (t['name'], t['ref_max'], t['cosine'], t['rel_l2'], t['max_abs_err']))
            lines.append('')
            continue
        algo = {0: 'fp32 MFMA', 1: '3 x bf16 pieces (default)', 2: '2 x fp16 pieces (opt-in)'}[r['algo']]
        lines += ['## %s  shape %s, products: %s' % (r['case'], 'x'.join(map(str, r['shape'])), algo), '',
                  'loss: engine %.9g, cpu32 %.9g%s; torch-CPU float32 oracle step %.1f s on %d threads' % (
                      r['loss'], r['loss_cpu32'], (', f64 %.9g' % r['loss_f64']) if r.get('loss_f64') is not None else '', r['cpu32_oracle_s'], r['threads']), '',
                  '| tensor | max abs ref | engine vs cpu32 | 1e-5(1+ref) | engine vs f64 | cpu32 vs f64 | within | rel | rel ok |', '|---|---|---|---|---|---|---|---|---|']
        for t in r['tensors']:
            within = '1e-5' if t['err_vs_cpu32'] <= t['bound_1e5'] else ('1e-5 (f64)' if t.get('err_vs_f64', 1e9) <= t['bound_1e5'] else 'FAIL')
            lines.append('| %s | %.3e | %.3e | %.3e | %s | %s | %s | %.1e | %s |' % (
                t['name'], t['ref_max'], t['err_vs_cpu32'], t['bound_1e5'],
                ('%.3e' % t['err_vs_f64']) if 'err_vs_f64' in t else '-', ('%.3e' % t['cpu32_vs_f64']) if 'cpu32_vs_f64' in t else '-', within,
                t.get('rel_err', float('nan')), 'yes' if t.get('rel_ok', True) else 'NO'))
        worst = max(r['tensors'], key=lambda t: t['err_vs_cpu32'] / t['bound_1e5'])
        lines += ['', 'worst tensor relative to the plain bound: `%s` at %.2f x' % (worst['name'], worst['err_vs_cpu32'] / worst['bound_1e5']), '']
    out = os.path.join(ROOT, 'profiles', '%s_parity.md' % tag)
    with open(out, 'w') as f:
        f.write('\n'.join(lines) + '\n')
    print('wrote', out)


if __name__ == '__main__':
    main(*sys.argv[1:])
