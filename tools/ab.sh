#!/bin/bash
# same-box A/B: cloudini_amd/lib/libcloudini_hip_A.so (baseline build) against the current library, alternating
for rep in 1 2; do
for lib in A cur; do
  if [ $lib = A ]; then export CLDN_HIP_LIB_OVERRIDE=$GRAFT_REPO_ROOT/cloudini_amd/lib/libcloudini_hip_A.so; else unset CLDN_HIP_LIB_OVERRIDE; fi
  for w in "c2" "c3 --clouds 16" "c4 --clouds 256"; do
    timeout 600 python bench.py --workload $w --steps 20 --warmup 3 --cpu-baseline-seconds 0 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('$lib', '$w'.ljust(18), round(d['value']), 'Mpts/s', {k: round(v,3) for k,v in d['device_ms_per_step'].items()})"
  done
done
done
