#!/bin/bash
# Diagnostic builds: scripts/build_variant.sh <name> <source.hip> "<extra -D flags>"  ->  csrc/libunet_hip_<name>.so
# (the other objects are the regular build's; load with UNET_HIP_LIB=<path>).  Never used by the product path.
set -e
cd "$(dirname "$0")/../semantic-segmentation-unet_amd/csrc"
name=$1; src=$2; flags=$3
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -fPIC -std=c++17 -Wall -Wno-unused-function $flags -c $src -o /tmp/variant_$name.o
objs=""
for o in *.o; do if [ "$o" != "${src%.hip}.o" ]; then objs="$objs $o"; fi; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o libunet_hip_$name.so $objs /tmp/variant_$name.o
echo built libunet_hip_$name.so
