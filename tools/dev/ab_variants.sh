#!/bin/bash
# on the GPU box: tools/dev/ab_variants.sh name1 name2 ...  (names of cloudini_amd/lib/variants/libcloudini_hip_<name>.so; "head" = the built library)
cd $GRAFT_REPO_ROOT
cp cloudini_amd/lib/libcloudini_hip.so /tmp/libcloudini_hip_head.so
for rep in 1 2 3; do
for v in "$@"; do
  if [ $v = head ]; then cp /tmp/libcloudini_hip_head.so cloudini_amd/lib/libcloudini_hip.so; else cp cloudini_amd/lib/variants/libcloudini_hip_$v.so cloudini_amd/lib/libcloudini_hip.so; fi
  timeout 600 python bench.py --steps 20 --warmup 5 --cpu-baseline-seconds 0 --e2e-seconds 0 --transcode-messages 0 --config-legs 0 ${BENCH_ARGS} 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('[$v]'.ljust(28), 'exact' if d.get('bit_exact') else 'WRONG', round(d['value']), 'Mpts/s median', round(d['repeats']['ms_per_step_median'],4), 'ms', {k: round(v,4) for k,v in d['device_ms_per_step'].items()})"
done
done
cp /tmp/libcloudini_hip_head.so cloudini_amd/lib/libcloudini_hip.so
