#!/bin/bash
# time composition of k_encode_floatn: CLDN_HIP_ABLATE bits 1 = no column output, 2 = no token emission, 4 = no ring flush, 8 = no loads, 16 = columns by direct stores (valid output)
for v in "$@"; do
  CLDN_HIP_ABLATE=$v timeout 300 python bench.py --steps 10 --warmup 2 --cpu-baseline-seconds 0 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('ablate', $v, 'k1_ms', round(686.598272/d['roofline']['achieved'],4), 'step_ms', round(d['ms_per_step'],4))"
done
