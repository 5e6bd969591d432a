#!/bin/bash
# parity on the encode cases, then bench kernel timings, for each CLDN_HIP_EXP variant given as arguments
for v in "$@"; do
  echo "=== EXP=$v"
  CLDN_HIP_EXP=$v timeout 900 python -m pytest tests/test_gpu_encode.py tests/test_golden.py -x -q -m gpu 2>&1 | tail -2
  CLDN_HIP_EXP=$v timeout 300 python bench.py --steps 10 --warmup 2 --cpu-baseline-seconds 0 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['achieved'], d.get('kernel_ms'))"
done
