#!/bin/bash
# Diagnostic: ablation variants of the fused Winograd weight-gradient kernel, same box.
#   0: product build   5: every DMA reads the same few KB (wrong results; removes fabric/HBM traffic)   6: no XCD renumbering
set -e
cd "$(dirname "$0")/.."
C=semantic-segmentation-unet_amd/csrc
for a in 0 5 6; do
  hipcc -O3 --offload-arch=gfx950 -fPIC -std=c++17 -DUNET_ABLATE=$a -shared -o /tmp/libunet_wgabl$a.so $C/conv_igemm.hip $C/conv_wgrad.hip $C/conv_direct.hip $C/winograd.hip $C/norm.hip $C/misc.hip 2>/dev/null
done
for a in 0 5 6 0; do
  echo "== ablate $a"
  UNET_HIP_LIB=/tmp/libunet_wgabl$a.so python scripts/bench_conv.py fwgrad 2>/dev/null | grep -E "fwgrad"
done
