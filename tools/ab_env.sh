#!/bin/bash
# same-box A/B of environment switches: every argument is one "VAR=val VAR2=val" set (use "-" for none); C2 bench each
for rep in 1 2; do
for envs in "$@"; do
  [ "$envs" = "-" ] && envs=""
  env $envs timeout 600 python bench.py --steps 20 --warmup 5 --cpu-baseline-seconds 0 --e2e-seconds 0 --transcode-messages 0 --config-legs 0 ${BENCH_ARGS} 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('[$envs]'.ljust(44), 'exact' if d.get('bit_exact') else 'WRONG', round(d['value']), 'Mpts/s median', round(d['repeats']['ms_per_step_median'],4), 'ms', {k: round(v,4) for k,v in d['device_ms_per_step'].items()})"
done
done
