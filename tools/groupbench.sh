#!/bin/bash
# chunk-group pipeline A/B: CLDN_HIP_GROUPS = 1 (off), 2, 4, 6, 8 and the automatic choice, per BASELINE config
for w in "c2" "c1 --clouds 256 --points 65536" "c4 --clouds 256" "c3 --clouds 16" "c5 --clouds 1 --points 10000000"; do
  for g in 1 2 4 8 auto; do
    if [ "$g" = auto ]; then unset CLDN_HIP_GROUPS; else export CLDN_HIP_GROUPS=$g; fi
    timeout 600 python bench.py --workload $w --steps 20 --warmup 3 --repeats 3 --cpu-baseline-seconds 0 --e2e-seconds 0 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('$w'.ljust(36), 'groups $g'.ljust(12), round(d['value']), 'Mpts/s', round(d['repeats']['ms_per_step_median'],3), 'ms (min', round(d['repeats']['ms_per_step_min'],3), ')', {k: round(v,3) for k,v in d['device_ms_per_step'].items()})"
  done
done
