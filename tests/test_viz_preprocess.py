"""cloudini_ros::applyVizLossyPreprocessing (ros_msg_utils.hpp:175-221, src/ros_msg_utils.cpp:249-341): NaN drop +
order-preserving voxel dedup. CPU: the oracle's restatement against the compiled reference; GPU: the HIP kernels
against the oracle, byte for byte."""
import numpy as np
import pytest

import cases
from cloudini_amd import synth
from cloudini_amd.schema import FieldType as F


def _viz_clouds():
    """(name, info, data, xyz_offset, resolution)"""
    rs = np.random.RandomState(5)
    out = []
    info, data = synth.lidar_xyzi(150_000, seed=3)
    pts = data.reshape(-1, 16).copy()
    pts[rs.randint(0, len(pts), 20000)] = pts[rs.randint(0, len(pts), 20000)]     # exact duplicates
    f = pts.view(np.float32).reshape(-1, 4)
    f[rs.randint(0, len(pts), 3000), rs.randint(0, 3, 3000)] = np.nan
    f[rs.randint(0, len(pts), 300), rs.randint(0, 3, 300)] = np.inf
    f[rs.randint(0, len(pts), 300), rs.randint(0, 3, 300)] = -np.inf
    f[1000:1100, 0] += np.float32(0.0004)      # same voxel as the neighbour row copied below, different bytes
    pts[1100:1200] = pts[1000:1100]
    f[1100:1200, 0] += np.float32(0.0002)
    out.append(("xyzi_dups_nans", info, pts.reshape(-1), 0, 0.001))
    # coarse voxels: most points collapse
    from cloudini_amd.schema import PointField
    info2 = info.copy(fields=[PointField(fl.name, fl.offset, fl.type, 0.25) if i < 3 else fl
                              for i, fl in enumerate(info.fields)])
    out.append(("xyzi_coarse", info2, data, 0, 0.25))
    # depth camera, 5 % NaN pixels, 32-byte points
    info3, data3 = synth.depthcam_xyzrgba(320, 240)
    out.append(("depthcam", info3, data3, 0, float(info3.fields[0].resolution)))
    # packed 18-byte points, triple at offset 0; values beyond the 21-bit key range and beyond int32 (wrap / indefinite)
    info4, data4 = synth.velodyne_xyzir(20000, seed=8)
    p4 = data4.reshape(-1, 18).copy()
    big = np.array([3000.0, -3000.0, 2.5e6, -2.5e6, 3e9, -3e9, 1e19, -1e19, 3.4e38], dtype=np.float32)
    for k, v in enumerate(big):
        p4[500 + 3 * k: 503 + 3 * k, 0:4] = np.frombuffer(np.float32(v).tobytes(), np.uint8)
    out.append(("packed18_extremes", info4, p4.reshape(-1), 0, 0.001))
    # triple not at offset 0, unaligned
    n = 30000
    fields = [("x", 3, F.FLOAT32, 0.01), ("y", 7, F.FLOAT32, 0.01), ("z", 11, F.FLOAT32, 0.01), ("i", 15, F.UINT16, None)]
    info5 = cases.make_info(fields, 19, n)
    data5 = cases.pack(info5, {"x": np.round(rs.uniform(-5, 5, n), 2).astype(np.float32),
                               "y": np.round(rs.uniform(-5, 5, n), 1).astype(np.float32),
                               "z": np.zeros(n, np.float32), "i": rs.randint(0, 65536, n).astype(np.uint16)}, n)
    out.append(("offset3_step19", info5, data5, 3, 0.01))
    out.append(("single_point", *synth.lidar_xyz(1)[:2], 0, 0.001))
    out.append(("empty", synth.lidar_xyz(1)[0], np.zeros(0, np.uint8), 0, 0.001))
    allnan = np.full(3 * 777, np.nan, dtype=np.float32).view(np.uint8)
    out.append(("all_nan", synth.lidar_xyz(777)[0], allnan, 0, 0.001))
    return out


VIZ = _viz_clouds()


@pytest.mark.parametrize("name,info,data,xyz_off,res", VIZ, ids=[v[0] for v in VIZ])
def test_oracle_matches_reference(oracle, reflib, name, info, data, xyz_off, res):
    want, res_after, width, height = reflib.viz_preprocess(info, data)
    got = oracle.viz_preprocess(data, info.point_step, xyz_off, res)
    assert np.array_equal(got, want)
    assert width == len(want) // info.point_step and height == 1 or data.size == 0


def _viz_random(seed):
    """Round 6: random clouds for the pre-filter -- point step, triple offset, resolution, cluster structure (so that voxels hold
    from one to thousands of points), NaN / Inf / huge values sprinkled in. (info, data, xyz_offset, resolution)"""
    import os
    rs = np.random.RandomState(seed)
    n = int(rs.choice([1, 63, 1025, 20_000, 70_001, 300_000]))
    off = int(rs.choice([0, 0, 1, 2, 4, 7]))
    extra = int(rs.choice([0, 0, 2, 4, 6, 20]))
    step = off + 12 + extra
    res = float(rs.choice([0.001, 0.01, 0.05, 0.25, 1.0]))
    kind = rs.randint(0, 4)
    if kind == 0:
        xyz = rs.uniform(-50, 50, (n, 3))
    elif kind == 1:                                              # a few hundred clusters, tight
        c = rs.uniform(-20, 20, (max(1, n // 200), 3))
        xyz = c[rs.randint(0, len(c), n)] + rs.normal(0, res * 0.7, (n, 3))
    elif kind == 2:                                              # a scan line: neighbours share voxels
        t = np.arange(n) * 1e-3
        xyz = np.stack([np.cos(t) * 10, np.sin(t) * 10, t * 0.01], axis=1)
    else:                                                        # a grid hit many times
        xyz = np.round(rs.uniform(-3, 3, (n, 3)) / (res * 2)) * (res * 2)
    xyz = xyz.astype(np.float32)
    k = max(1, n // 100)
    if rs.rand() < 0.6:
        xyz[rs.randint(0, n, k), rs.randint(0, 3, k)] = np.nan
    if rs.rand() < 0.3:
        xyz[rs.randint(0, n, k), rs.randint(0, 3, k)] = rs.choice([np.inf, -np.inf, 3e9, -3e9, 2.5e6, 1e19], k).astype(np.float32)
    fields = [("x", off, F.FLOAT32, res), ("y", off + 4, F.FLOAT32, res), ("z", off + 8, F.FLOAT32, res)]
    cols = {"x": xyz[:, 0], "y": xyz[:, 1], "z": xyz[:, 2]}
    if extra >= 2:
        fields.append(("i", off + 12, F.UINT16, None))
        cols["i"] = rs.randint(0, 65536, n).astype(np.uint16)
    info = cases.make_info(fields, step, n)
    return info, cases.pack(info, cols, n), off, res


@pytest.mark.parametrize("seed", list(range(100, 160)))
def test_oracle_matches_reference_on_random_clouds(oracle, reflib, seed):
    info, data, off, res = _viz_random(seed)
    want, _res_after, _w, _h = reflib.viz_preprocess(info, data)
    assert np.array_equal(oracle.viz_preprocess(data, info.point_step, off, res), want), seed


@pytest.mark.gpu
@pytest.mark.parametrize("seed", list(range(100, 160)) + list(range(20_000, 20_000 + int(__import__("os").environ.get("CLDN_FUZZ_EXTRA", "0")) // 50)))
def test_gpu_matches_oracle_on_random_clouds(oracle, seed):
    """... host buffers, and device-resident ones at odd addresses."""
    import torch
    from cloudini_amd import native
    info, data, off, res = _viz_random(seed)
    step = info.point_step
    n = data.size // step
    want = oracle.viz_preprocess(data, step, off, res)
    codec = native.Codec(native.Plan(synth.lidar_xyz(1)[0]))
    got = codec.viz_preprocess_host(data, step, off, res)
    assert got.size == want.size and np.array_equal(got, want), (seed, got.size // step, want.size // step)
    dev = torch.device("cuda", 0)
    mi, mo = seed % 4, seed // 4 % 4
    d_in = torch.zeros(data.size + 8, dtype=torch.uint8, device=dev)
    d_in[mi:mi + data.size] = torch.from_numpy(data).to(dev)
    d_out = torch.zeros(data.size + 8, dtype=torch.uint8, device=dev)
    kept = codec.viz_preprocess_device(d_in.data_ptr() + mi, n, step, off, res, d_out.data_ptr() + mo, data.size)
    codec.synchronize()
    assert kept * step == want.size and np.array_equal(d_out[mo:mo + kept * step].cpu().numpy(), want), (seed, "device resident")
    codec.close()


def test_reference_no_triple_is_a_noop(reflib):
    """No geometry triple (offsets not consecutive / resolutions differ / fewer than 3 fields) -> untouched."""
    n = 100
    rs = np.random.RandomState(2)
    fields = [("x", 0, F.FLOAT32, 0.01), ("y", 8, F.FLOAT32, 0.01), ("z", 4, F.FLOAT32, 0.01)]
    info = cases.make_info(fields, 12, n)
    data = rs.randint(0, 255, 12 * n).astype(np.uint8)
    got, _res, width, _h = reflib.viz_preprocess(info, data)
    assert np.array_equal(got, data) and width == n


@pytest.mark.gpu
@pytest.mark.parametrize("name,info,data,xyz_off,res", VIZ, ids=[v[0] for v in VIZ])
def test_gpu_matches_oracle(oracle, name, info, data, xyz_off, res):
    from cloudini_amd import native
    codec = native.Codec(native.Plan(synth.lidar_xyz(1)[0]))   # the codec only lends device, stream, workspace
    got = codec.viz_preprocess_host(data, info.point_step, xyz_off, res)
    want = oracle.viz_preprocess(data, info.point_step, xyz_off, res)
    assert got.size == want.size, (got.size // info.point_step, want.size // info.point_step)
    assert np.array_equal(got, want)
    codec.close()


@pytest.mark.gpu
def test_gpu_large_cloud_and_errors(oracle):
    from cloudini_amd import native
    info, data = synth.lidar_xyzi(2_000_000, seed=12)
    codec = native.Codec(native.Plan(info))
    got = codec.viz_preprocess_host(data, 16, 0, 0.05)
    want = oracle.viz_preprocess(data, 16, 0, 0.05)
    assert np.array_equal(got, want) and 0 < got.size < data.size
    for bad in ((16, 8, 0.001), (16, 0, 0.0), (16, 0, float("nan")), (16, 0, -1.0)):
        with pytest.raises(native.CloudiniHipError):
            codec.viz_preprocess_host(data[:1600], *bad)
    codec.close()


@pytest.mark.gpu
def test_host_mirror_matches_the_reference_function():
    """cloudini_ros::applyVizLossyPreprocessing of the host mirror (gate, shape update, FLOAT64 -> 1 us) against
    the reference's outputs frozen by the CPU suite on the same inputs (tests/golden/viz_golden.npz)."""
    import os
    from cloudini_amd import api
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "viz_golden.npz"), allow_pickle=False)
    for name, info, data in _schema_gate_cases():
        got, res, w, h = api.applyVizLossyPreprocessing(info, data)
        assert np.array_equal(got, g[name + "/out"]), name
        want_res = g[name + "/res"]
        got_res = np.array([np.nan if r is None else r for r in res], dtype=np.float32)
        assert np.array_equal(got_res, want_res, equal_nan=True), name
        assert (w, h) == tuple(int(x) for x in g[name + "/shape"]), name


def _schema_gate_cases():
    """Schemas around the gate of the function: with a triple, without one, FLOAT64 fields with / without resolution."""
    rs = np.random.RandomState(9)
    n = 5000

    def cloud(fields, step):
        info = cases.make_info(fields, step, n)
        cols = {}
        for name, _off, typ, _res in fields:
            if typ == F.FLOAT32:
                cols[name] = np.round(rs.uniform(-3, 3, n), 1).astype(np.float32)
            elif typ == F.FLOAT64:
                cols[name] = rs.uniform(0, 1e9, n)
            else:
                cols[name] = rs.randint(0, 200, n).astype(np.uint16)
        first_float = [f[0] for f in fields if f[2] == F.FLOAT32][0]
        cols[first_float][::50] = np.nan
        return info, cases.pack(info, cols, n)

    out = []
    out.append(("triple_f64_stamp", *cloud([("x", 0, F.FLOAT32, 0.1), ("y", 4, F.FLOAT32, 0.1), ("z", 8, F.FLOAT32, 0.1),
                                             ("t", 16, F.FLOAT64, None), ("s", 24, F.FLOAT64, 0.5), ("i", 12, F.UINT16, None)], 32)))
    out.append(("res_differs", *cloud([("x", 0, F.FLOAT32, 0.1), ("y", 4, F.FLOAT32, 0.2), ("z", 8, F.FLOAT32, 0.1),
                                        ("t", 16, F.FLOAT64, None)], 24)))
    out.append(("offsets_not_consecutive", *cloud([("x", 0, F.FLOAT32, 0.1), ("y", 8, F.FLOAT32, 0.1), ("z", 4, F.FLOAT32, 0.1)], 12)))
    out.append(("no_resolution", *cloud([("x", 0, F.FLOAT32, None), ("y", 4, F.FLOAT32, None), ("z", 8, F.FLOAT32, None)], 12)))
    out.append(("two_fields", *cloud([("x", 0, F.FLOAT32, 0.1), ("y", 4, F.FLOAT32, 0.1)], 8)))
    out.append(("int_first", *cloud([("i", 0, F.UINT16, None), ("x", 4, F.FLOAT32, 0.1), ("y", 8, F.FLOAT32, 0.1), ("z", 12, F.FLOAT32, 0.1)], 16)))
    out.append(("negative_resolution", *cloud([("x", 0, F.FLOAT32, -0.1), ("y", 4, F.FLOAT32, -0.1), ("z", 8, F.FLOAT32, -0.1)], 12)))
    return out


def test_freeze_reference_outputs_for_the_gpu_box(reflib, tmp_path):
    """Runs where /root/reference exists: the reference's outputs for the gate cases must equal the committed golden
    file (regenerate with tests/golden/make_viz_golden.py if the cases change)."""
    import os
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "viz_golden.npz"), allow_pickle=False)
    for name, info, data in _schema_gate_cases():
        out, res, w, h = reflib.viz_preprocess(info, data)
        assert np.array_equal(out, g[name + "/out"]), name
        assert np.array_equal(np.array(res, dtype=np.float32), g[name + "/res"], equal_nan=True), name
        assert (w, h) == tuple(int(x) for x in g[name + "/shape"]), name
