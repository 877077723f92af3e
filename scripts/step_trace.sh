#!/bin/bash
# Per-kernel timeline of one training step: scripts/step_trace.sh <bf16|f32> <out prefix> [extra bench flags]   (run from the repo root under gpurun)
# writes <prefix>_timeline.txt (every kernel of the last step in start order: start / end / duration / queue) -- scripts/step_timeline.py on a rocpd database
set -e
d=$1; pre=$2; shift 2
root=$(pwd); cd /tmp; export TMPDIR=/tmp
extra=""; if [ $d = bf16 ]; then extra="--dtype bf16 --channels 3 --classes 4"; fi
rm -rf /tmp/trace_$d
rocprofv3 --kernel-trace --output-format rocpd -d /tmp/trace_$d -- python3 $root/bench.py $extra --steps 4 --warmup 2 --no-extra --no-cpu-baseline --no-kernel-events "$@" > /tmp/trace_$d.log 2>&1
db=$(find /tmp/trace_$d -name "*.db" | head -1)
python3 $root/scripts/step_timeline.py $db > $root/${pre}_timeline.txt
tail -16 $root/${pre}_timeline.txt
