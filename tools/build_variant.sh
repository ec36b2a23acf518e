#!/bin/bash
# dev: build a library variant for same-box A/B runs.  usage: tools/build_variant.sh <git-rev|WORK> <out.so> [extra hipcc flags]
set -e
cd "$(dirname "$0")/.."
REV=$1; OUT=$2; shift 2
T=/tmp/eld_variant_$$; rm -rf $T; mkdir -p $T/eld_amd/csrc $T/include
if [ "$REV" = "WORK" ]; then cp eld_amd/csrc/*.hip eld_amd/csrc/*.h $T/eld_amd/csrc/; cp include/*.h $T/include/
else git archive $REV eld_amd/csrc include | tar x -C $T; fi
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt $@"
for s in $T/eld_amd/csrc/*.hip; do /opt/rocm/bin/hipcc $F -c $s -o ${s%.hip}.o & done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT $T/eld_amd/csrc/*.o
rm -rf $T; ls -la $OUT
