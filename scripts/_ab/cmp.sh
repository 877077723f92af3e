for lib in - semantic-segmentation-unet_amd/csrc/libunet_hip_ctold.so; do
  if [ "$lib" = "-" ]; then unset UNET_HIP_LIB; else export UNET_HIP_LIB=$PWD/$lib; fi
  python bench.py --dtype bf16 --channels 3 --classes 4 --steps 48 --warmup 8 --no-extra --no-cpu-baseline > gpurun_out/cmp_$(basename $lib .so).json 2>/dev/null
  python - <<PY
import json
d=json.loads([l for l in open("gpurun_out/cmp_$(basename $lib .so).json") if l.startswith("{")][-1])
print("$lib", d["ms_per_step"], {k: round(v["ms_per_step"],3) for k,v in d["kernels"].items() if "convt" in k or k in ("bn_apply","bn_bwd")})
PY
done
