#!/usr/bin/env python3
"""CPU, needs /root/reference (oracle/_ref): the C oracle's decoder against the compiled reference on damaged streams --
same accept / reject decision, same bytes when accepted. The damages and schema families of
tests/test_gpu_fuzz.py::test_corrupted_streams_decode_like_the_oracle (the GPU is compared with the ORACLE there: this
run pins the checker itself) over a seed range given on the command line: first_seed count [processes]."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))


def run(args):
    first, count = args
    from oracle import binding
    from cloudini_amd import synth
    import test_gpu_fuzz as T
    orc, ref = binding.Oracle(), binding.RefLib()
    n_acc = n_rej = 0
    bad = []
    for seed in range(first, first + count):
        rs = np.random.RandomState(seed)
        pick = rs.randint(0, 5)
        if pick == 0:
            info, data = synth.lidar_xyzi(int(rs.choice([500, 5000, 40000])), seed=seed)
        elif pick == 1:
            info, data = synth.lidar_xyz(int(rs.choice([300, 33000])), seed=seed)
        elif pick == 2:
            info, data = synth.velodyne_xyzir(int(rs.choice([1000, 20000])), seed=seed)
        elif pick == 3:
            info, data = synth.depthcam_xyzrgba(64, 48, seed=seed)
        else:
            info, data = T._random_case(1000 + seed % 100)
        n = data.size // info.point_step
        s = orc.encode_stage1(info, data).copy()
        if len(s) < 8:
            continue
        kind = rs.randint(0, 4)
        if kind == 0:
            for _ in range(int(rs.randint(1, 4))):
                s[rs.randint(0, len(s))] ^= np.uint8(1 << rs.randint(0, 8))
        elif kind == 1:
            s = s[: rs.randint(1, len(s))]
        elif kind == 2:
            pos = rs.randint(4, len(s))
            s = np.concatenate([s[:pos], rs.randint(0, 256, int(rs.randint(1, 4))).astype(np.uint8), s[pos:]])
        else:
            pos = rs.randint(4, len(s) - 1)
            s = np.concatenate([s[:pos], s[pos + 1:]])
        try:
            a = orc.decode_stage1(info, s, n, fill=0xE1)
        except Exception:
            a = None
        try:
            b = ref.decode_noheader(info.copy(width=n, height=1), s, fill=0xE1)
        except Exception:
            b = None
        if (a is None) != (b is None) or (a is not None and not np.array_equal(a, b)):
            bad.append(seed)
        elif a is None:
            n_rej += 1
        else:
            n_acc += 1
    return n_acc, n_rej, bad


if __name__ == "__main__":
    from multiprocessing import Pool
    first, count = int(sys.argv[1]), int(sys.argv[2])
    procs = int(sys.argv[3]) if len(sys.argv) > 3 else 8
    per = (count + procs - 1) // procs
    jobs = [(first + k * per, min(per, first + count - (first + k * per))) for k in range(procs) if first + k * per < first + count]
    with Pool(procs) as pool:
        res = pool.map(run, jobs)
    acc = sum(r[0] for r in res); rej = sum(r[1] for r in res); bad = sum((r[2] for r in res), [])
    print(f"seeds {first}..{first + count - 1}: oracle == reference on {acc + rej} damaged streams ({acc} accepted with equal bytes, {rej} rejected by both); disagreements: {bad}")
