#!/usr/bin/env python3
"""Turn two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; counter_collection.csv each) of `bench.py` into
profiles/traffic.json, which bench.py reports as roofline.traffic.
  bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024   -- FETCH_SIZE/WRITE_SIZE are in KB; on gfx950 FETCH_SIZE counts half of a
  wide (16 B/lane) coalesced read stream (MI355X_MICROARCH.md, HBM section), every read in these kernels is 16 B/lane.
The hash of the library that ran the passes (eld_build_info "src=", written as build_info.txt by tools/gpu_profile_set.sh into the directory above the
CSVs) is recorded as library_src_hash / library_src_hash_bf16: bench.py reports these bytes only for a library built from the same sources.
Usage: traffic_from_pmc.py <fetch_counter_collection.csv> <write_counter_collection.csv> <out.json> [frames per pass = bench.py --batch, default 8] [bf16]
With a fifth argument `bf16` the passes are those of `bench.py --precision bf16`: the kernels and the per-pass total are MERGED into an existing
<out.json> under `kernels_bf16` / `unet_conv_bytes_per_pass_bf16` (bench.py reports it as alt_bf16.roofline.traffic)."""
import collections
import csv
import json
import sys


def load(path, counter):
    tot = collections.defaultdict(float)
    calls = collections.Counter()
    grid = collections.defaultdict(float)
    seen = set()
    for r in csv.DictReader(open(path)):
        if r['Counter_Name'] != counter:
            continue
        k = r['Kernel_Name']
        tot[k] += float(r['Counter_Value'])
        if r['Dispatch_Id'] not in seen:
            seen.add(r['Dispatch_Id'])
            calls[k] += 1
            grid[k] += float(r['Grid_Size'])
    return tot, calls, grid


def src_hash_of(csv_path):
    """src=<hash> of the build_info.txt one or two directories above the counter CSV; None when the run did not record one."""
    import os
    d = os.path.dirname(os.path.abspath(csv_path))
    for up in (d, os.path.dirname(d), os.path.dirname(os.path.dirname(d))):
        p = os.path.join(up, 'build_info.txt')
        if os.path.exists(p):
            t = open(p).read()
            return t.rsplit('src=', 1)[1].split()[0].strip() if 'src=' in t else None
    return None


def main():
    f, fc, fg = load(sys.argv[1], 'FETCH_SIZE')
    w, wc, wg = load(sys.argv[2], 'WRITE_SIZE')
    out = {'source': 'rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) of `bench.py --steps 1 --warmup 1 --no-cpu-baseline`; '
                     'bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 (gfx950 FETCH_SIZE x2 correction for 16 B/lane streams)', 'kernels': {}}
    for k in f:
        rd, wr = 2.0 * f[k] * 1024.0, w.get(k, 0.0) * 1024.0
        out['kernels'][k[:80]] = {'calls': fc[k], 'read_bytes': rd, 'write_bytes': wr, 'bytes_per_launch': (rd + wr) / max(fc[k], 1)}
    noise = [k for k in f if k.startswith('void noise_kernel')]
    if noise:
        px = sum(fg[k] / 256.0 * 4096.0 for k in noise)           # 256 threads per workgroup, 4096 pixels per workgroup
        b = sum(2.0 * f[k] * 1024.0 + w.get(k, 0.0) * 1024.0 for k in noise)
        out['sampler_bytes_per_pixel'] = b / px
    passes = sum(fc[k] for k in f if 'head_fwd_kernel' in k or 'head_train_kernel' in k)      # one head launch per forward pass (fused training head or plain)
    conv = [k for k in f if any(s in k for s in ('conv_igemm_kernel', 'conv_x3', 'wgrad_kernel', 'wgrad8_kernel', 'conv_first', 'wgrad_reduce'))]
    if passes and conv:
        out['unet_passes'] = passes
        out['frames_per_pass'] = int(sys.argv[4]) if len(sys.argv) > 4 else 8
        out['unet_conv_bytes_per_pass'] = sum(2.0 * f[k] * 1024.0 + w.get(k, 0.0) * 1024.0 for k in conv) / passes
    if len(sys.argv) > 5 and sys.argv[5] == 'bf16':
        passes = sum(fc[k] for k in f if 'head_fwd_kernel' in k or 'head_train_kernel' in k)
        conv = [k for k in f if any(s_ in k for s_ in ('conv_igemm_kernel', 'conv_bf', 'wgrad8_kernel', 'wgrad8d_kernel', 'wgrad_kernel', 'conv_first', 'wgrad_reduce'))]
        try:
            base = json.load(open(sys.argv[3]))
        except Exception:
            base = {}
        base['kernels_bf16'] = out['kernels']
        base['library_src_hash_bf16'] = src_hash_of(sys.argv[1])
        base['source_bf16'] = out['source'].replace('--no-cpu-baseline', '--precision bf16 --no-cpu-baseline')
        if passes and conv:
            base['unet_passes_bf16'] = passes
            base['unet_conv_bytes_per_pass_bf16'] = sum(2.0 * f[k] * 1024.0 + w.get(k, 0.0) * 1024.0 for k in conv) / passes
        json.dump(base, open(sys.argv[3], 'w'), indent=1)
        print(json.dumps({k: v for k, v in base.items() if not k.startswith('kernels')}, indent=1))
        return
    out['library_src_hash'] = src_hash_of(sys.argv[1])
    # optional: --sq <counter_collection.csv of the SQ pass of the same command>: the sampler's issued VALU instructions (its real limiter)
    if '--sq' in sys.argv:
        sq = sys.argv[sys.argv.index('--sq') + 1]
        v, vc, vg = load(sq, 'SQ_INSTS_VALU')
        nz = [k for k in v if k.startswith('void noise_kernel')]
        if nz:
            px = sum(vg[k] / 256.0 * 4096.0 for k in nz)         # as above: 4096 pixels per 256-thread workgroup
            out['sampler_valu_lane_ops_per_pixel'] = 64.0 * sum(v[k] for k in nz) / px
            out['sampler_valu_source'] = 'rocprofv3 --pmc SQ_INSTS_VALU (wave instructions x 64 lanes / pixels) of the noise_kernel launches of the same bench.py command'
    json.dump(out, open(sys.argv[3], 'w'), indent=1)
    print(json.dumps({k: v for k, v in out.items() if k != 'kernels'}, indent=1))


if __name__ == '__main__':
    main()
