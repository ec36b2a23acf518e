#!/bin/bash
# register / LDS / spill figures of every kernel in one .hip file (gfx950 device asm): tools/kregs.sh eld_amd/csrc/conv_x3.hip
f=$1; out=/tmp/$(basename $f .hip).s
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -I include --cuda-device-only -S $f -o $out 2>/dev/null
python3 - $out <<'PY'
import re,sys
t=open(sys.argv[1]).read()
for blk in re.findall(r'- \.agpr_count:.*?\.wavefront_size', t, re.S):
    g=lambda k:(re.search(r'\.%s:\s+(\S+)'%k, blk) or [None,'?'])[1]
    print('%-70s vgpr %3s agpr %3s sgpr %3s spill %s lds %s'%(g('name')[:70], g('vgpr_count'), g('agpr_count'), g('sgpr_count'), g('vgpr_spill_count'), g('group_segment_fixed_size')))
PY
