#!/usr/bin/env python3
"""CPU: the oracle's decoder against the compiled reference on the damaged streams of DIVERSE schemas that
tests/test_gpu_fuzz.py::test_damaged_streams_of_diverse_schemas_decode_like_the_oracle runs (same generator, same seeds):
tools/dev/oracle_vs_ref_damaged_diverse.py <first seed> <count>"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_fuzz as T
from oracle.binding import Oracle, RefLib
o, r = Oracle(), RefLib()
first, count = int(sys.argv[1]), int(sys.argv[2])
t0 = time.time(); acc = rej = skipped = 0
for seed in range(first, first + count):
    rs, info, data = T._damaged_case(seed)
    n = data.size // info.point_step
    s = o.encode_stage1(info, data)
    if len(s) < 8:
        skipped += 1; continue
    s = T._damage(rs, s)
    try: a = o.decode_stage1(info, s, n, fill=0xE1)
    except Exception: a = None
    try: b = r.decode_noheader(info.copy(width=n, height=1), s, fill=0xE1)
    except Exception: b = None
    assert (a is None) == (b is None), ("decision", seed)
    if a is not None:
        assert np.array_equal(a, b), ("bytes", seed); acc += 1
    else: rej += 1
print(f"seeds {first}..{first + count - 1}: oracle == reference on all; accepted by both {acc}, rejected by both {rej}, empty {skipped}; {time.time() - t0:.0f} s")
