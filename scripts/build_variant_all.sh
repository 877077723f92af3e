#!/bin/bash
# Diagnostic build of the WHOLE library with extra flags:  scripts/build_variant_all.sh <name> "<-D flags>"  ->  csrc/libunet_hip_<name>.so
# (load with UNET_HIP_LIB=<path>).  Never used by the product path.
set -e
root="$(cd "$(dirname "$0")/.." && pwd)"
name=$1; flags=$2
cd "$root/semantic-segmentation-unet_amd/csrc"
srcs=$(python3 -c "import sys; sys.path.insert(0, '$root/semantic-segmentation-unet_amd'); import _build; print(' '.join(_build.SOURCES))")
mkdir -p /tmp/variant_$name
echo $srcs | tr ' ' '\n' | xargs -P 8 -I{} sh -c "/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -fPIC -std=c++17 -Wall -Wno-unused-function $flags -c {} -o /tmp/variant_$name/{}.o 2>/dev/null"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o libunet_hip_$name.so /tmp/variant_$name/*.o
echo built libunet_hip_$name.so
