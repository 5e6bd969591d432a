"""The N>1 path on CPU: 2 gloo ranks shard a ragged batch by cloud and exchange the encoded sizes. The encode
function is injected; here it is the CPU oracle (the GPU tests inject the HIP codec), so what is being tested is
the sharding / size-exchange logic of cloudini_amd/sharding.py, exactly as bench.py and a batch transcoder use it."""
import os
import sys

import numpy as np
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, sizes, q):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist

    from cloudini_amd import sharding, synth
    from oracle.binding import Oracle

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    oracle = Oracle()
    clouds, info = [], None
    for k, n in enumerate(sizes):
        info, data = synth.lidar_xyzi(n, seed=50 + k)
        clouds.append(data)

    def encode_fn(mine):
        return [oracle.encode_stage1(info, c) for c in mine]

    owned, streams, all_sizes, offsets, total = sharding.encode_sharded(clouds, encode_fn, rank, world)
    dist.barrier()
    q.put((rank, owned, [len(s) for s in streams], all_sizes.tolist(), offsets.tolist(), total))
    dist.destroy_process_group()


def test_two_rank_sharding_and_size_exchange():
    sizes = [1000, 40000, 0, 5000, 33000]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, sizes, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = sorted(q.get(timeout=240) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, own0, len0, all0, off0, tot0), (r1, own1, len1, all1, off1, tot1) = results
    assert own0 == [0, 2, 4] and own1 == [1, 3]
    assert all0 == all1 and off0 == off1 and tot0 == tot1  # every rank knows the whole layout
    assert [all0[k] for k in own0] == len0 and [all0[k] for k in own1] == len1
    # the layout is the single-process one
    sys.path.insert(0, ROOT)
    from cloudini_amd import synth
    from oracle.binding import Oracle
    o = Oracle()
    want = [len(o.encode_stage1(*synth.lidar_xyzi(n, seed=50 + k))) for k, n in enumerate(sizes)]
    assert all0 == want and tot0 == sum(want) and off0 == list(np.cumsum([0] + want[:-1]))
