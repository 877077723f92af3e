#!/bin/bash
# Diagnostic: ablation variants of the fused Winograd kernel (wrong results by construction), same box.
set -e
cd "$(dirname "$0")/.."
C=semantic-segmentation-unet_amd/csrc
for a in 0 1 2 3; do
  hipcc -O3 --offload-arch=gfx950 -fPIC -std=c++17 -DUNET_ABLATE=$a -shared -o /tmp/libunet_wabl$a.so $C/conv_igemm.hip $C/conv_wgrad.hip $C/conv_direct.hip $C/winograd.hip $C/norm.hip $C/misc.hip 2>/dev/null
done
for a in 0 1 2 3 0; do
  echo "== ablate $a (1: no U DMA in loop, 2: no transform in loop, 3: neither nor D DMA)"
  UNET_HIP_LIB=/tmp/libunet_wabl$a.so python scripts/bench_conv.py ffwd 2>/dev/null | grep -E "^1b|^2b|^4b|^bott_b|TOTAL"
done
