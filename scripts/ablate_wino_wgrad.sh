#!/bin/bash
# Diagnostic: variants of the fused Winograd weight-gradient kernel on the same box.
#   product build | 6: no XCD renumbering | 7: s_memtime phase timeline (scripts/wgrad_timeline.py)
set -e
cd "$(dirname "$0")/.."
C=semantic-segmentation-unet_amd/csrc
bash scripts/build_variant.sh wg6 winograd.hip "-DUNET_ABLATE=6" > /dev/null
bash scripts/build_variant.sh wg7 winograd.hip "-DUNET_ABLATE=7" > /dev/null
echo "== product"; python scripts/bench_conv.py fwgrad 2>/dev/null | grep -E "fwgrad"
echo "== no XCD renumbering"; UNET_HIP_LIB=$C/libunet_hip_wg6.so python scripts/bench_conv.py fwgrad 2>/dev/null | grep -E "fwgrad"
echo "== phase timeline (cycles per chunk, workgroup 0 wave 0)"; UNET_HIP_LIB=$C/libunet_hip_wg7.so python scripts/wgrad_timeline.py 2>/dev/null
