#!/bin/bash
# Diagnostic: package power and shader clock (rocm-smi, read-only) sampled while bench.py runs one workload.
# usage (GPU box, repo root): bash scripts/power_probe.sh "<bench.py flags>" <label>
flags=$1; label=$2
python3 bench.py $flags --steps ${STEPS:-400} --warmup 10 --no-extra --no-cpu-baseline --no-kernel-events > /tmp/pp_$label.json 2>/dev/null &
pid=$!
sleep 14
for i in 1 2 3 4 5 6; do
  /opt/rocm/bin/rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Power|sclk|mclk" | tr '\n' ' ' | sed 's/  */ /g'; echo
  sleep 1
done
wait $pid
python3 -c "
import json
d=json.loads([l for l in open('/tmp/pp_$label.json') if l.startswith('{')][-1]); print('$label', d['value'], 'images/s', d['ms_per_step'], 'ms')"
