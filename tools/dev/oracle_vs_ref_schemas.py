#!/usr/bin/env python3
"""CPU, needs /root/reference (oracle/_ref): the C oracle against the compiled reference on the random schemas of
tests/test_gpu_fuzz.py (_random_case: field types, offsets with padding, resolutions, encoding options, wire versions drawn
at random) -- encoded bytes equal, decoded bytes equal. The GPU campaigns compare with the ORACLE on these schemas: this run
pins the checker on them. Arguments: first_seed count [processes] [wide]."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))


def run(args):
    first, count, wide = args
    from oracle import binding
    import test_gpu_fuzz as T
    orc, ref = binding.Oracle(), binding.RefLib()
    ok, bad = 0, []
    for seed in range(first, first + count):
        info, data = T._random_case(seed, wide=wide)
        n = data.size // info.point_step
        a = orc.encode_stage1(info, data)
        b = ref.encode_stage1(info.copy(width=n, height=1), data)
        if not np.array_equal(a, b):
            bad.append((seed, "encode"))
            continue
        da = orc.decode_stage1(info, a, n, fill=0xC3)
        db = ref.decode_noheader(info.copy(width=n, height=1), a, fill=0xC3)
        if not np.array_equal(da, db):
            bad.append((seed, "decode"))
            continue
        ok += 1
    return ok, bad


if __name__ == "__main__":
    from multiprocessing import Pool
    first, count = int(sys.argv[1]), int(sys.argv[2])
    procs = int(sys.argv[3]) if len(sys.argv) > 3 else 8
    wide = len(sys.argv) > 4 and sys.argv[4] == "wide"
    per = (count + procs * 8 - 1) // (procs * 8)
    jobs = [(s, min(per, first + count - s), wide) for s in range(first, first + count, per)]
    with Pool(procs) as pool:
        res = pool.map(run, jobs)
    ok = sum(r[0] for r in res); bad = sum((r[1] for r in res), [])
    print(f"{'wide ' if wide else ''}seeds {first}..{first + count - 1}: oracle == reference (stage-1 bytes and decoded bytes) on {ok} random schemas; disagreements: {bad}")
