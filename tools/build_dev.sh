#!/bin/bash
# dev build of the library with the ablation switches compiled in (ELD_DEV_TOOLS=1) -> tools/probe/libeld_dev.so (never loaded by the package)
set -e
cd "$(dirname "$0")/.."
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -DELD_DEV_TOOLS=1"
mkdir -p /tmp/eld_dev
for s in eld_amd/csrc/*.hip; do /opt/rocm/bin/hipcc $F -c $s -o /tmp/eld_dev/$(basename $s .hip).o & done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/probe/libeld_dev.so /tmp/eld_dev/*.o
ls -la tools/probe/libeld_dev.so
