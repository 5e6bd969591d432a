#!/bin/bash
# like ab_variants.sh, printing the decode leg and the config legs' decode times: tools/dev/ab_variants_dec.sh name1 name2 ...
cd $GRAFT_REPO_ROOT
cp cloudini_amd/lib/libcloudini_hip.so /tmp/libcloudini_hip_head.so
for rep in 1 2; do
for v in "$@"; do
  if [ $v = head ]; then cp /tmp/libcloudini_hip_head.so cloudini_amd/lib/libcloudini_hip.so; else cp cloudini_amd/lib/variants/libcloudini_hip_$v.so cloudini_amd/lib/libcloudini_hip.so; fi
  timeout 600 python bench.py --steps 20 --warmup 5 --cpu-baseline-seconds 0 --e2e-seconds 0 --transcode-messages 0 ${BENCH_ARGS} 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
cf = d.get('configs') or {}
print('[$v]'.ljust(12), 'exact' if d.get('bit_exact') else 'WRONG', 'enc', round(d['ms_per_step'],4), 'dec', round(d['decode']['ms_per_step'],4), 'dec0', round(d['decode']['fill_zero_ms_per_step'],4), 'kern', round(d['decode']['roofline']['kernel_ms'],4), {k: (round(v.get('decode_ms',0),4), v.get('bit_exact')) for k,v in cf.items()})"
done
done
cp /tmp/libcloudini_hip_head.so cloudini_amd/lib/libcloudini_hip.so
