#!/bin/bash
# Diagnostic builds: scripts/build_variant.sh <name> <source.hip> "<extra -D flags>"  ->  csrc/libunet_hip_<name>.so
# (load with UNET_HIP_LIB=<path>; _lib.py still checks its ABI version).  Never used by the product path.
# The other objects are the regular build's: the regular library is (re)built first, so every object corresponds to the sources in the
# tree (content stamps, _build.py) and no stale *.o from an earlier tree can be linked in.
set -e
root="$(cd "$(dirname "$0")/.." && pwd)"
python3 "$root/semantic-segmentation-unet_amd/_build.py" > /dev/null
cd "$root/semantic-segmentation-unet_amd/csrc"
name=$1; src=$2; flags=$3
extra=$(python3 -c "import sys; sys.path.insert(0, '$root/semantic-segmentation-unet_amd'); import _build; print(' '.join(_build.EXTRA_FLAGS.get('$src', [])))")
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -fPIC -std=c++17 -Wall -Wno-unused-function $extra $flags -c $src -o /tmp/variant_$name.o
objs=""
for s in $(python3 -c "import sys; sys.path.insert(0, '$root/semantic-segmentation-unet_amd'); import _build; print(' '.join(_build.SOURCES))"); do
    if [ "$s" != "$src" ]; then objs="$objs ${s%.hip}.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o libunet_hip_$name.so $objs /tmp/variant_$name.o
echo built libunet_hip_$name.so
