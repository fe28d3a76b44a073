#!/bin/bash
# What the FD float bin-pair analysis kernel spends its time on: builds of the library with parts of the kernel compiled out
# (scripts/f32_kernel_parts.patch adds the SDFT_X_* switches -- wrong results, timing only; the product tree stays untouched)
# side by side with the product, then scripts/f32_variant_probe.py on the GPU box:
#   bash scripts/f32_kernel_parts.sh          (here: builds into sdft_amd/lib/probe/)
#   gpurun -- 'python scripts/f32_variant_probe.py product=sdft_amd/lib/libsdft_hip.so no_dpp=sdft_amd/lib/probe/libsdft_hip_nodpp.so ...'
set -e
cd "$(dirname "$0")/.."
git apply scripts/f32_kernel_parts.patch
trap 'git apply -R scripts/f32_kernel_parts.patch; python -m sdft_amd.build --force > /dev/null 2>&1' EXIT
mkdir -p sdft_amd/lib/probe
for v in NODPP NOSTORE NOBAR NOLDS "NOSTORE -DSDFT_X_NOBAR -DSDFT_X_NOLDS -DSDFT_X_NODPP"; do
  name=$(echo $v | tr -d ' ' | tr 'A-Z' 'a-z' | sed 's/-dsdft_x_/_/g')
  SDFT_HIP_EXTRA_FLAGS="-DSDFT_X_$v" python -m sdft_amd.build --force > /dev/null 2>&1
  cp sdft_amd/lib/libsdft_hip.so sdft_amd/lib/probe/libsdft_hip_$name.so
  echo built $name
done
