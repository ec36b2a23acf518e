#!/bin/bash
# dev build of the library with the ablation switches compiled in (ELD_DEV_TOOLS=1) -> tools/probe/libeld_dev.so (never loaded by the package)
cd "$(dirname "$0")/.."
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -DELD_DEV_TOOLS=1"
rm -rf /tmp/eld_dev; mkdir -p /tmp/eld_dev
pids=()
for s in eld_amd/csrc/*.hip; do /opt/rocm/bin/hipcc $F -c $s -o /tmp/eld_dev/$(basename $s .hip).o & pids+=($!); done
fail=0
for p in "${pids[@]}"; do wait $p || fail=1; done
if [ $fail -ne 0 ]; then echo "build_dev: a compile FAILED (no library written)"; rm -f tools/probe/libeld_dev.so; exit 1; fi
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/probe/libeld_dev.so /tmp/eld_dev/*.o || exit 1
ls -la tools/probe/libeld_dev.so
