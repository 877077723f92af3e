#!/bin/bash
# Collect the round's judged profiles on the GPU box:  scripts/collect_profiles.sh <tag> [stages, default 123]   (from the repo root under gpurun;
# the three stages together take ~15 min, more than one gpurun call allows: run "1", "2", "3" in separate calls)
#   gpurun_out/<tag>_bench_{f32,bf16}_kernel_stats.txt            rocprofv3 --kernel-trace --stats of bench.py (two-stream backward)
#   gpurun_out/<tag>_bench_{f32,bf16}_exclusive_kernel_stats.txt  the same with --no-overlap (every duration exclusive)
#   gpurun_out/<tag>_{x6_fwd,x6_dgrad,wino_wgrad,bf16_fwd,bf16_dgrad,bf16_wgrad}_pmc_traffic.json   FETCH_SIZE / WRITE_SIZE in two separate --pmc
#                                                                  passes (config 2 on the default BF16x6 route / config 4)
#   gpurun_out/<tag>_wino_{fwd,dgrad}_pmc_traffic.json             the same for --fp32-matrix native (fp32-MFMA forward / data gradient)
#   gpurun_out/<tag>_config5_{x6_fwd,x6_dgrad,wino_wgrad}_pmc_traffic.json  at BASELINE config 5's per-GPU workload (1024x1024x3, 6 classes, batch 2)
#   gpurun_out/<tag>_layer_table_{f32,f32native,bf16}.txt          per-layer ms / executed TFLOP/s / mfma_busy of the 3x3 families inside the step
#   gpurun_out/<tag>_overlap_standin.txt                           stand-in collective under the backward pass (scripts/overlap_probe.py)
# Only --kernel-trace / --stats / --pmc are used (never combined with other trace domains); python3 itself follows `--`.
set -e
set -o pipefail
tag=$1; root=$(pwd); out=$root/gpurun_out; mkdir -p $out
cd /tmp; export TMPDIR=/tmp
BF="--dtype bf16 --channels 3 --classes 4"
stages=${2:-123}
A32=7981465600; A16=3990732800    # algorithmic bytes per step of the 17 MFMA 3x3 layers: input + output activations + weights, 4 B (fp32) or 2 B (bf16) per element
if [[ $stages == *1* ]]; then
for d in f32 f32native bf16; do
  extra=""; if [ $d = bf16 ]; then extra=$BF; fi; if [ $d = f32native ]; then extra="--fp32-matrix native"; fi
  for mode in "" "--no-overlap"; do
    if [ $d = f32native ] && [ -z "$mode" ]; then continue; fi
    name=${tag}_bench_${d}; what="two-stream backward"
    if [ -n "$mode" ]; then name=${name}_exclusive; what="SINGLE-STREAM backward (--no-overlap: every duration exclusive)"; fi
    rm -rf /tmp/prof_$name
    rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$name -- python3 $root/bench.py $extra --steps 6 --warmup 2 --no-extra --no-cpu-baseline --no-kernel-events $mode > /tmp/prof_$name.log 2>&1
    ips=$(grep '^{' /tmp/prof_$name.log | python3 -c "import sys,json; print(json.loads(sys.stdin.readline())['value'])")
    csv=$(find /tmp/prof_$name -name "*kernel_stats.csv" | head -1)
    python3 $root/scripts/kernel_stats_summary.py $csv 8 "$tag, $d, $what: rocprofv3 --kernel-trace --stats -- python3 bench.py $extra --steps 6 --warmup 2 --no-extra --no-cpu-baseline --no-kernel-events $mode; $ips images/s under the profiler" > $out/${name}_kernel_stats.txt
    echo "$name: $ips images/s"
  done
done
fi
if [[ $stages == *2* ]]; then
# PMC traffic: one whole (last) step of a single-stream run per counter
for d in f32 f32native bf16; do
  extra=""; if [ $d = bf16 ]; then extra=$BF; fi; if [ $d = f32native ]; then extra="--fp32-matrix native"; fi
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/pmc_${d}_$c
    rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pmc_${d}_$c -- python3 $root/bench.py $extra --steps 2 --warmup 1 --no-extra --no-cpu-baseline --no-kernel-events --no-overlap > /tmp/pmc_${d}_$c.log 2>&1
    echo "pmc $d $c done"
  done
done
ff=$(find /tmp/pmc_f32_FETCH_SIZE -name "*counter_collection.csv" | head -1); fw=$(find /tmp/pmc_f32_WRITE_SIZE -name "*counter_collection.csv" | head -1)
nf=$(find /tmp/pmc_f32native_FETCH_SIZE -name "*counter_collection.csv" | head -1); nw=$(find /tmp/pmc_f32native_WRITE_SIZE -name "*counter_collection.csv" | head -1)
bf=$(find /tmp/pmc_bf16_FETCH_SIZE -name "*counter_collection.csv" | head -1); bw=$(find /tmp/pmc_bf16_WRITE_SIZE -name "*counter_collection.csv" | head -1)
cd $root
python3 scripts/pmc_traffic.py $ff $fw wino_x6_stream_stats_kernel 17 "512x512x1/2 classes/batch 8/f32 (BF16x6)" $A32 > $out/${tag}_x6_fwd_pmc_traffic.json
python3 scripts/pmc_traffic.py $ff $fw wino_x6_stream_bnbwd_kernel,wino_x6_stream_kernel 17 "512x512x1/2 classes/batch 8/f32 (BF16x6)" $A32 > $out/${tag}_x6_dgrad_pmc_traffic.json
python3 scripts/pmc_traffic.py $nf $nw wino_fused_stream_stats_kernel 17 "512x512x1/2 classes/batch 8/f32 (native fp32 MFMA)" $A32 > $out/${tag}_wino_fwd_pmc_traffic.json
python3 scripts/pmc_traffic.py $nf $nw wino_fused_stream_bnbwd_kernel,wino_fused_stream_kernel 17 "512x512x1/2 classes/batch 8/f32 (native fp32 MFMA)" $A32 > $out/${tag}_wino_dgrad_pmc_traffic.json
python3 scripts/pmc_traffic.py $ff $fw wino_wgrad_fused_kernel 17 "512x512x1/2 classes/batch 8/f32" $A32 > $out/${tag}_wino_wgrad_pmc_traffic.json
python3 scripts/pmc_traffic.py $bf $bw conv_bf16_stream_stats_kernel,conv_bf16_stream_in_stats_kernel,conv_bf16_stats_kernel 17 "512x512x3/4 classes/batch 8/bf16" $A16 1 > $out/${tag}_bf16_fwd_pmc_traffic.json
python3 scripts/pmc_traffic.py $bf $bw conv_bf16_stream_bnbwd_kernel,conv_bf16_stream_in_bnbwd_kernel,conv_bf16_bnbwd_kernel,conv_bf16_stream_kernel_,conv_bf16_stream_in_kernel_,conv_bf16_kernel_ 17 "512x512x3/4 classes/batch 8/bf16" $A16 1 > $out/${tag}_bf16_dgrad_pmc_traffic.json
python3 scripts/pmc_traffic.py $bf $bw ::wgrad_bf16_dma_kernel,::wgrad_bf16_kernel 17 "512x512x3/4 classes/batch 8/bf16" $A16 > $out/${tag}_bf16_wgrad_pmc_traffic.json
fi
if [[ $stages == *3* ]]; then
# BASELINE config 5 (1024x1024x3, 6 classes, batch 2): same pixel count per step as config 2, so the same algorithmic bytes for the 17 layers
C5="--size 1024 --channels 3 --classes 6 --batch 2"
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_c5_$c
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pmc_c5_$c -- python3 $root/bench.py $C5 --steps 2 --warmup 1 --no-extra --no-cpu-baseline --no-kernel-events --no-overlap > /tmp/pmc_c5_$c.log 2>&1
  echo "pmc config5 $c done"
done
cf=$(find /tmp/pmc_c5_FETCH_SIZE -name "*counter_collection.csv" | head -1); cw=$(find /tmp/pmc_c5_WRITE_SIZE -name "*counter_collection.csv" | head -1)
cd $root
python3 scripts/pmc_traffic.py $cf $cw wino_x6_stream_stats_kernel 17 "1024x1024x3/6 classes/batch 2/f32 (BF16x6)" $A32 > $out/${tag}_config5_x6_fwd_pmc_traffic.json
python3 scripts/pmc_traffic.py $cf $cw wino_x6_stream_bnbwd_kernel,wino_x6_stream_kernel 17 "1024x1024x3/6 classes/batch 2/f32 (BF16x6)" $A32 > $out/${tag}_config5_x6_dgrad_pmc_traffic.json
python3 scripts/pmc_traffic.py $cf $cw wino_wgrad_fused_kernel 17 "1024x1024x3/6 classes/batch 2/f32" $A32 > $out/${tag}_config5_wino_wgrad_pmc_traffic.json
# per-layer table (ms, executed TFLOP/s, mfma_busy) of the three families inside one single-stream step, both precisions
cd /tmp
for d in f32 f32native bf16; do
  extra=""; ck="1 2"; if [ $d = bf16 ]; then extra=$BF; ck="3 4"; fi; if [ $d = f32native ]; then extra="--fp32-matrix native"; fi
  rm -rf /tmp/pmc_busy_$d
  rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY --output-format csv -d /tmp/pmc_busy_$d -- python3 $root/bench.py $extra --steps 2 --warmup 1 --no-extra --no-cpu-baseline --no-kernel-events --no-overlap > /tmp/pmc_busy_$d.log 2>&1
  bb=$(find /tmp/pmc_busy_$d -name "*counter_collection.csv" | head -1)
  (cd $root && python3 scripts/layer_table.py $bb $d $ck > $out/${tag}_layer_table_$d.txt)
  echo "layer table $d done"
done
cd $root
python3 scripts/overlap_probe.py 2>/dev/null > $out/${tag}_overlap_standin.txt
fi
echo collected stages $stages
