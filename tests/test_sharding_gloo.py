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


# ---- one large cloud split by chunk ranges (second row) ------------------------------------------

def _worker_chunks(rank, world, port, n_points, q):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist

    from cloudini_amd import sharding, synth
    from oracle.binding import Oracle

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    oracle = Oracle()
    info, data = synth.lidar_xyzi(n_points, seed=77)
    n_modes = oracle.adaptive_field_count(info)

    def probe_fn(head):
        return oracle.encode_stage1(info, head, return_modes=True)[1]

    def encode_part_fn(part, modes):
        return oracle.encode_stage1_continued(info, part, modes)

    part, offset, total, modes = sharding.encode_cloud_sharded(data, info.point_step, n_modes, probe_fn, encode_part_fn,
                                                               rank, world)
    dist.barrier()
    q.put((rank, part.tobytes(), offset, total, modes.tolist()))
    dist.destroy_process_group()


def test_chunk_ranges():
    from cloudini_amd.sharding import shard_chunks
    assert shard_chunks(100000, 2, 0) == (0, 65536) and shard_chunks(100000, 2, 1) == (65536, 34464)
    assert shard_chunks(32768, 2, 0) == (0, 32768) and shard_chunks(32768, 2, 1) == (32768, 0)
    assert shard_chunks(0, 4, 3) == (0, 0)
    for n in (1, 32767, 32769, 1_000_000, 10_000_000):
        for w in (1, 2, 3, 8):
            parts = [shard_chunks(n, w, r) for r in range(w)]
            assert sum(c for _, c in parts) == n
            assert all(p % 32768 == 0 for p, _ in parts)
            assert all(parts[r][0] + parts[r][1] == parts[r + 1][0] or parts[r + 1][1] == 0 for r in range(w - 1))


def test_two_rank_one_cloud_by_chunk_ranges():
    """Rank 0 probes the modes on the head of the cloud, both ranks encode their chunk range with those modes; the
    concatenation must be the single-process stream (the second range alone would pick another mode for the
    intensity column if it probed by itself -- the palette of its own first 4096 values differs)."""
    n_points = 150_000  # 5 chunks -> 3 + 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker_chunks, args=(r, 2, port, n_points, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = sorted(q.get(timeout=240) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (_, b0, off0, tot0, m0), (_, b1, off1, tot1, m1) = results
    sys.path.insert(0, ROOT)
    from cloudini_amd import synth
    from oracle.binding import Oracle
    info, data = synth.lidar_xyzi(n_points, seed=77)
    whole, modes = Oracle().encode_stage1(info, data, return_modes=True)
    assert m0 == m1 == modes.tolist()
    assert off0 == 0 and off1 == len(b0) and tot0 == tot1 == len(whole)
    assert b0 + b1 == whole.tobytes()


def test_oracle_continued_matches_whole_cloud_for_every_mode():
    """orc_encode_stage1_continued against the tail of a whole-cloud encode, for schemas committing each of the
    four adaptive modes (the reference-pinned oracle is the only source of truth for the whole cloud)."""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import cases
    from oracle.binding import Oracle
    o = Oracle()
    seen = set()
    for name, info, data, _modes in cases.reference_int_sequences():
        n = data.size // info.point_step
        if n <= 32768:
            continue
        whole, modes = o.encode_stage1(info, data, return_modes=True)
        head = o.encode_stage1(info, data[: 32768 * info.point_step])
        tail = o.encode_stage1_continued(info, data[32768 * info.point_step:], modes)
        assert head.tobytes() + tail.tobytes() == whole.tobytes(), name
        seen.update(modes.tolist())
    for kind in ("grows_u16", "single_value"):
        info, data = cases.palette_stress(kind)
        whole, modes = o.encode_stage1(info, data, return_modes=True)
        head = o.encode_stage1(info, data[: 32768 * info.point_step])
        tail = o.encode_stage1_continued(info, data[32768 * info.point_step:], modes)
        assert head.tobytes() + tail.tobytes() == whole.tobytes(), kind
        seen.update(modes.tolist())
    assert len(seen) >= 2
