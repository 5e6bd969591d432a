#!/bin/bash
# one bench line per BASELINE config (device-resident encode), condensed
for w in "c1 --clouds 256 --points 65536" "c2" "c3 --clouds 16" "c4 --clouds 256" "c5 --clouds 1 --points 10000000" "c2 --clouds 1" "c2 --clouds 4"; do
  timeout 600 python bench.py --workload $w --steps 10 --warmup 2 --cpu-baseline-seconds 0 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('$w'.ljust(40), round(d['value']), 'Mpts/s', round(d['ms_per_step'],3), 'ms', {k: round(v,3) for k,v in d['device_ms_per_step'].items()}, 'roof', round(d['roofline']['frac'],3), 'B/pt', round(d['stage1_bytes_per_point'],2))"
done
