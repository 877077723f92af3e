#!/bin/bash
# Diagnostic: build igemm ablation variants (wrong results by construction) and time them on the same box.
set -e
cd "$(dirname "$0")/.."
C=semantic-segmentation-unet_amd/csrc
for a in 0 1 2 3; do
  hipcc -O3 --offload-arch=gfx950 -fPIC -std=c++17 -DUNET_ABLATE=$a -shared -o /tmp/libunet_abl$a.so $C/conv_igemm.hip $C/conv_wgrad.hip $C/conv_direct.hip $C/norm.hip $C/misc.hip 2>/dev/null
done
for a in 0 1 2 3 0; do
  echo "== ablate $a"
  UNET_HIP_LIB=/tmp/libunet_abl$a.so python scripts/bench_conv.py fwd 2>/dev/null | grep -E "^1b|^3b|^4b|^dec_1a|TOTAL"
done
