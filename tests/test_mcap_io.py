"""MCAP container I/O of the batch transcoder (include/cloudini_amd/mcap_io.hpp).

The reference reads and writes bags through the upstream mcap library (tools/src/mcap_converter.cpp:32-57, :141-300), which
this image does not have -- nor any sample bag. What can be pinned here:
  * the C++ writer and reader against each other (three chunk compressions, many small chunks);
  * both against tests/mcap_py.py, a second statement of the published record layouts in another language: Python reads what
    C++ wrote (every record, the summary section and its offsets included), C++ reads what Python wrote (chunked or not);
  * on the GPU: a bag with point clouds and other topics goes through `cloudini_batch_transcode in.mcap out.mcap` -- every
    converted message equals the reference's converter byte for byte, everything else arrives untouched and in file order,
    the schema of the cloud topics is swapped -- and back again.
Against the mcap library itself the parity is UNPINNED."""
import os
import struct
import subprocess

import numpy as np
import pytest

import mcap_py

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "cloudini_amd", "lib")


def _build(tmp_path, name):
    exe = str(tmp_path / name)
    subprocess.run(["g++", "-std=c++17", "-O1", "-Wall", "-pthread", "-I" + os.path.join(ROOT, "include"),
                    os.path.join(ROOT, "tests", "cpp", name + ".cpp"), os.path.join(LIB, "libcloudini_amd.so"),
                    os.path.join(LIB, "libcloudini_hip.so"), "-Wl,-rpath," + LIB, "-Wl,-rpath,/opt/rocm/lib", "-o", exe], check=True)
    return exe


def _payload(i):
    n = 17 + (i * 37) % 400
    return bytes((((i * 131 + k * 7) & 0xffffffff) >> (k & 3)) & 0xff for k in range(n))


def _fnv(b):
    h = 1469598103934665603
    for x in b:
        h = ((h ^ x) * 1099511628211) & 0xffffffffffffffff
    return h


def test_cpp_writer_and_reader_round_trip_and_python_reads_the_file(tmp_path):
    exe = _build(tmp_path, "mcap_roundtrip")
    r = subprocess.run([exe, str(tmp_path)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "all checks passed" in r.stdout, r.stdout + r.stderr
    f = mcap_py.read(str(tmp_path / "rt_none.mcap"))
    assert f["header"] == ("ros2", "cloudini_amd")
    assert set(f["schemas"]) == {1, 7} and f["schemas"][1][0] == "sensor_msgs/msg/PointCloud2" and f["schemas"][7][2] == b"string data"
    assert f["channels"][3] == (1, "/lidar/points", "cdr", [("offered_qos_profiles", "x")])
    assert f["channels"][9] == (1, "/depth/points", "cdr", [("k", "v"), ("k2", "")])
    assert f["metadata"] == [("rosbag2", [("ROS_DISTRO", "jazzy")])]
    assert len(f["messages"]) == 60
    for i, (ch, seq, lt, pt, data) in enumerate(f["messages"]):
        assert (ch, seq, lt, pt) == ((3, 4, 9)[i % 3], i, 1000 + 10 * i, 999 + 10 * i) and data == _payload(i)
    assert len(f["chunks"]) > 10  # 700-byte chunks
    # the summary section says where things are, and it is right
    s = f["summary"]
    assert s["schemas"] == f["schemas"] and s["channels"] == f["channels"]
    st = s["statistics"]
    assert (st["messages"], st["schemas"], st["channels"], st["metadata"], st["chunks"]) == (60, 2, 3, 1, len(f["chunks"]))
    assert (st["t0"], st["t1"]) == (1000, 1590) and st["counts"] == {3: 20, 4: 20, 9: 20}
    assert [(c[2], c[3]) for c in s["chunk_index"]] == [(c[0], c[1]) for c in f["chunks"]]           # offset, record length
    assert [(c[0], c[1], c[6], c[7]) for c in s["chunk_index"]] == [(c[2], c[3], c[6], c[4]) for c in f["chunks"]]  # times, sizes
    summary_recs = [x for x in f["records"] if x[3] == "summary" and x[0] not in (mcap_py.DATA_END,)]
    first_summary = min(x[1] for x in summary_recs if x[0] not in (mcap_py.FOOTER,))
    first_offset = min(x[1] for x in summary_recs if x[0] == mcap_py.SUMMARY_OFFSET)
    assert f["footer"] == (first_summary, first_offset, 0)
    for g, start, length in s["offsets"]:
        inside = [x for x in f["records"] if x[3] == "summary" and start <= x[1] < start + length]
        assert inside and all(x[0] == g for x in inside) and sum(x[2] for x in inside) == length
    # round 5: MessageIndex records behind every chunk -- one per channel with messages in it, every (log_time, offset) pointing
    # at a Message record of that channel inside the uncompressed chunk -- and their places in the ChunkIndex records
    raw = open(str(tmp_path / "rt_none.mcap"), "rb").read()
    mi = f["message_index"]
    chunks = f["chunks"]
    seen = 0
    for k, (coff, clen, _t0, _t1, usize, _comp, n2) in enumerate(chunks):
        nxt = chunks[k + 1][0] if k + 1 < len(chunks) else None
        mine = [m for m in mi if m[0] >= coff + clen and (nxt is None or m[0] < nxt)]
        body0 = coff + clen - n2                      # where the chunk's (uncompressed) records begin in the file
        assert s["chunk_index_message_offsets"][k] == {m[2]: m[0] for m in mine}
        assert s["chunk_index"][k][4] == sum(m[1] for m in mine)     # message_index_length
        assert [m[2] for m in mine] == sorted({m[2] for m in mine})
        for _off, _len, ch, entries in mine:
            for lt, o in entries:
                assert raw[body0 + o] == mcap_py.MESSAGE
                got_ch, _seq, got_lt = struct.unpack_from("<HIQ", raw, body0 + o + 9)
                assert (got_ch, got_lt) == (ch, lt) and o < usize
                seen += 1
    assert seen == 60
    assert not os.path.exists(str(tmp_path / "rt_none.mcap.partial"))


def test_cpp_writer_that_is_not_closed_leaves_no_file(tmp_path):
    """only close() finalizes; a writer destroyed on the way (an exception in the conversion) must not leave a
    well-formed bag that silently misses its tail. A reader given a chunk that claims 4 GiB must refuse it, not allocate it."""
    exe = _build(tmp_path, "mcap_roundtrip")
    r = subprocess.run([exe, str(tmp_path), "abandon"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "abandoned writer left nothing" in r.stdout and "oversized chunk refused" in r.stdout, r.stdout + r.stderr


@pytest.mark.parametrize("chunked", [None, 7])
def test_cpp_reader_takes_what_python_wrote(tmp_path, chunked):
    exe = _build(tmp_path, "mcap_dump")
    schemas = [(1, "sensor_msgs/msg/PointCloud2", "ros2msg", b"abc"), (2, "x/msg/Y", "ros2msg", b"")]
    channels = [(5, 1, "/a", "cdr", [("q", "1")]), (6, 2, "/b", "cdr", []), (8, 0, "/schemaless", "json", [])]
    msgs = [((5, 6, 8)[i % 3], i, 50 + i, 40 + i, _payload(i)) for i in range(25)]
    path = str(tmp_path / "py.mcap")
    mcap_py.write(path, "ros2", schemas, channels, msgs, metadata=[("m", [("k", "v")])], chunk_messages=chunked)
    r = subprocess.run([exe, path], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0, r.stdout + r.stderr
    lines = r.stdout.strip().split("\n")
    assert lines[0] == "header ros2"
    assert [l for l in lines if l.startswith("schema")] == ["schema 1 sensor_msgs/msg/PointCloud2 ros2msg 3", "schema 2 x/msg/Y ros2msg 0"]
    assert [l for l in lines if l.startswith("channel")] == ["channel 5 1 /a cdr 1", "channel 6 2 /b cdr 0", "channel 8 0 /schemaless json 0"]
    assert [l for l in lines if l.startswith("metadata")] == ["metadata m 1"]
    got = [l for l in lines if l.startswith("message")]
    assert got == [f"message {c} {q} {lt} {pt} {len(d)} {_fnv(d)}" for c, q, lt, pt, d in msgs]
    # McapStream (record by record, a chunk at a time) reads the same -- also with the declarations in the middle of the file
    late = str(tmp_path / "late.mcap")
    mcap_py.write(late, "ros2", schemas, channels, msgs, metadata=[("m", [("k", "v")])], chunk_messages=chunked, declare_late=True)
    for f in (path, late):
        rs = subprocess.run([exe, f, "stream"], capture_output=True, text=True, timeout=60)
        assert rs.returncode == 0 and rs.stdout == subprocess.run([exe, f], capture_output=True, text=True, timeout=60).stdout, rs.stdout + rs.stderr
    assert rs.stdout == r.stdout
    # a file cut short is an error, not a crash
    cut = str(tmp_path / "cut.mcap")
    open(cut, "wb").write(open(path, "rb").read()[:-20])
    for mode in ([], ["stream"]):
        r = subprocess.run([exe, cut] + mode, capture_output=True, text=True, timeout=60)
        assert r.returncode == 1 and r.stdout.startswith("error MCAP:")


@pytest.mark.gpu
def test_a_bag_goes_through_the_tool_and_back(tmp_path, reflib):
    from cloudini_amd import synth
    from test_host_api import _cdr_pointcloud2
    tool = os.path.join(LIB, "cloudini_batch_transcode")
    clouds = []
    for k, n in enumerate([40000, 1, 70000, 0, 32768, 5000]):
        info, data = synth.lidar_xyzi(n, seed=10 + k)
        clouds.append(_cdr_pointcloud2(info, data, stamp=(1700000000 + k, 1000 * k)))
    info, data = synth.velodyne_xyzir(130048, seed=3)
    clouds.append(_cdr_pointcloud2(info, data, frame_id="velodyne"))
    pc2 = "sensor_msgs/msg/PointCloud2"
    schemas = [(1, pc2, "ros2msg", b"whatever the recorder wrote"), (2, "std_msgs/msg/String", "ros2msg", b"string data")]
    channels = [(1, 1, "/lidar", "cdr", []), (2, 2, "/chatter", "cdr", [("k", "v")]), (3, 1, "/velodyne", "cdr", [])]
    msgs, t = [], 1000
    for k, c in enumerate(clouds):
        msgs.append((2, 2 * k, t, t, b"hello %d" % k)); t += 5
        msgs.append((3 if k == 6 else 1, 2 * k + 1, t, t - 1, c.tobytes())); t += 5
    msgs.append((2, 99, t, t, b"bye"))
    src, enc, dec = str(tmp_path / "in.mcap"), str(tmp_path / "enc.mcap"), str(tmp_path / "dec.mcap")
    mcap_py.write(src, "ros2", schemas, channels, msgs, metadata=[("rosbag2", [("a", "b")])], chunk_messages=4)
    r = subprocess.run([tool, src, enc, "--resolution", "0.001", "--compression", "lz4", "--mcap-compression", "none", "--batch", "3"],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    f = mcap_py.read(enc)
    assert f["header"][0] == "ros2" and f["metadata"] == [("rosbag2", [("a", "b")])]
    assert f["schemas"][1][0] == "point_cloud_interfaces/msg/CompressedPointCloud2" and b"compressed_data" in f["schemas"][1][2]
    assert f["schemas"][2] == ("std_msgs/msg/String", "ros2msg", b"string data")
    assert f["channels"] == {c[0]: (c[1], c[2], c[3], c[4]) for c in channels}
    assert len(f["messages"]) == len(msgs)
    for (ch, seq, lt, pt, data), (ch0, seq0, lt0, pt0, data0) in zip(f["messages"], msgs):
        assert (ch, seq, lt, pt) == (ch0, seq0, lt0, pt0)
        if ch0 == 2:
            assert data == data0
        else:
            want = reflib.ros_compress(np.frombuffer(data0, dtype=np.uint8), 0.001, 1)  # LZ4
            assert data == want.tobytes()
    # the way back: zstd chunks this time (read by the C++ reader only), then once more uncompressed for the comparison
    r = subprocess.run([tool, enc, dec, "--decode", "--mcap-compression", "none"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    g = mcap_py.read(dec)
    assert g["schemas"][1][0] == pc2 and len(g["messages"]) == len(msgs)
    for (ch, seq, lt, pt, data), (ch0, seq0, lt0, pt0, data0), (_, _, _, _, e) in zip(g["messages"], msgs, f["messages"]):
        assert (ch, seq, lt, pt) == (ch0, seq0, lt0, pt0)
        if ch0 == 2:
            assert data == data0
        else:
            assert data == reflib.ros_decompress(np.frombuffer(e, dtype=np.uint8), len(data0) + 4096).tobytes()
    # compressed containers: zstd out, read again by the tool (lz4 out), sizes sane
    z, l = str(tmp_path / "z.mcap"), str(tmp_path / "l.mcap")
    assert subprocess.run([tool, src, z, "--mcap-compression", "zstd"], capture_output=True, timeout=300).returncode == 0
    assert subprocess.run([tool, z, l, "--decode", "--mcap-compression", "lz4"], capture_output=True, timeout=300).returncode == 0
    assert os.path.getsize(z) < os.path.getsize(src) and os.path.getsize(l) > os.path.getsize(z)


@pytest.mark.gpu
def test_a_bag_is_streamed_not_loaded(tmp_path, reflib):
    """transcodeMcap reads through McapStream (round 5): a chunk at a time, the messages that are not point clouds written
    as they come. Channels declared in the middle of the file; order and bytes as in the input; and the resident memory of
    the tool does not follow the size of the bag."""
    from cloudini_amd import synth
    from test_host_api import _cdr_pointcloud2
    tool = os.path.join(LIB, "cloudini_batch_transcode")
    pc2 = "sensor_msgs/msg/PointCloud2"
    schemas = [(1, pc2, "ros2msg", b"x"), (2, "sensor_msgs/msg/Image", "ros2msg", b"y")]
    channels = [(1, 2, "/camera", "cdr", []), (2, 1, "/lidar", "cdr", []), (3, 2, "/camera2", "cdr", []), (4, 1, "/lidar2", "cdr", [])]
    clouds = []
    for k in range(6):
        info, data = synth.lidar_xyzi(20000 + 1000 * k, seed=30 + k)
        clouds.append(_cdr_pointcloud2(info, data, stamp=(1700000000 + k, k)).tobytes())

    def bag(path, image_bytes, images_between):
        rs = np.random.RandomState(5)
        msgs, t = [], 10
        for k, c in enumerate(clouds):
            for j in range(images_between):
                msgs.append((1 if k < 3 else 3, len(msgs), t, t, rs.bytes(16) + bytes(image_bytes - 16))); t += 1
            msgs.append((2 if k < 4 else 4, len(msgs), t, t, c)); t += 1
        for j in range(images_between):
            msgs.append((3, len(msgs), t, t, rs.bytes(16) + bytes(image_bytes - 16))); t += 1
        mcap_py.write(path, "ros2", schemas, channels, msgs, chunk_messages=3, declare_late=True)
        return msgs

    def run(src, dst):
        # the tool reports its own high-water mark (CLDN_DEBUG_MEM: VmHWM of /proc/self/status, in kB). getrusage's
        # ru_maxrss of a child is no use here: it starts from the resident size of the process that forked it.
        r = subprocess.run([tool, src, dst, "--resolution", "0.001", "--compression", "zstd", "--mcap-compression", "none", "--batch", "2"],
                           capture_output=True, text=True, timeout=600, env=dict(os.environ, CLDN_DEBUG_MEM="1"))
        assert r.returncode == 0, r.stdout + r.stderr
        hwm = [int(line.split()[1]) for line in r.stderr.splitlines() if line.startswith("VmHWM")]
        assert len(hwm) == 1
        import json
        return hwm[0], json.loads(r.stdout.strip().splitlines()[-1])

    small, mid, big = str(tmp_path / "small.mcap"), str(tmp_path / "mid.mcap"), str(tmp_path / "big.mcap")
    msgs = bag(small, 64, 2)
    run(small, str(tmp_path / "small_out.mcap"))
    f = mcap_py.read(str(tmp_path / "small_out.mcap"))
    assert f["channels"] == {c[0]: (c[1], c[2], c[3], c[4]) for c in channels}
    assert f["schemas"][1][0] == "point_cloud_interfaces/msg/CompressedPointCloud2" and f["schemas"][2][0] == "sensor_msgs/msg/Image"
    assert len(f["messages"]) == len(msgs)
    for (ch, seq, lt, pt, data), (ch0, seq0, lt0, pt0, data0) in zip(f["messages"], msgs):
        assert (ch, seq, lt, pt) == (ch0, seq0, lt0, pt0)
        if ch0 in (1, 3):
            assert data == data0
        else:
            assert data == reflib.ros_compress(np.frombuffer(data0, dtype=np.uint8), 0.001, 2).tobytes()  # ZSTD
    # the same clouds between 7 x 40 and 7 x 80 images of 1 MiB: bags of 290 and 590 MB. What the tool holds at most is 64 MB
    # of copied-through messages plus what lies between two clouds; a reader that keeps the file image and its chunks
    # would need the size of the bag, twice. (The runtime's own footprint -- page-locked staging, the HIP libraries --
    # is in both runs: the difference is what follows the bag.)
    bag(mid, 1 << 20, 40)
    rss_mid, st_mid = run(mid, str(tmp_path / "mid_out.mcap"))
    os.remove(str(tmp_path / "mid_out.mcap"))
    bag(big, 1 << 20, 80)
    rss_big, st_big = run(big, str(tmp_path / "big_out.mcap"))
    grown = os.path.getsize(big) - os.path.getsize(mid)
    assert grown > 250e6 and os.path.getsize(str(tmp_path / "big_out.mcap")) > 500e6
    assert st_big["messages"] == 6 + 7 * 80 and st_big["converted"] == 6
    assert st_mid["peak_held_bytes"] <= (64 + 40 + 1) << 20 and st_big["peak_held_bytes"] <= (64 + 80 + 1) << 20
    assert (rss_big - rss_mid) * 1024 < 0.25 * grown, (rss_mid, rss_big, grown)


@pytest.mark.gpu
def test_a_cloud_in_a_partial_batch_does_not_hold_the_rest_of_the_bag(tmp_path, reflib):
    """A point cloud that sits in a batch which is not full (default --batch 64: three clouds never fill one) followed by far
    more than 64 MB of other messages -- a lidar topic that ends early. The source must hand its partial batch on when the
    held-back messages pass the limit (MessageSource::more), not buffer everything up to the end of the bag."""
    import json
    from cloudini_amd import synth
    from test_host_api import _cdr_pointcloud2
    tool = os.path.join(LIB, "cloudini_batch_transcode")
    pc2 = "sensor_msgs/msg/PointCloud2"
    schemas = [(1, pc2, "ros2msg", b"x"), (2, "sensor_msgs/msg/Image", "ros2msg", b"y")]
    channels = [(1, 2, "/camera", "cdr", []), (2, 1, "/lidar", "cdr", [])]
    rs = np.random.RandomState(11)
    msgs, t = [], 10
    clouds = []
    for k in range(3):
        info, data = synth.lidar_xyzi(15000 + 500 * k, seed=70 + k)
        clouds.append(_cdr_pointcloud2(info, data, stamp=(1700000100 + k, k)).tobytes())
    image = 1 << 20
    for k, n_images in enumerate((2, 150, 90)):  # cloud 0, 2 images, cloud 1, 150 MB of images, cloud 2, 90 MB of images, end
        msgs.append((2, len(msgs), t, t, clouds[k])); t += 1
        for j in range(n_images):
            msgs.append((1, len(msgs), t, t, rs.bytes(16) + bytes(image - 16))); t += 1
    src, dst = str(tmp_path / "sparse.mcap"), str(tmp_path / "sparse_out.mcap")
    mcap_py.write(src, "ros2", schemas, channels, msgs, chunk_messages=4)
    r = subprocess.run([tool, src, dst, "--resolution", "0.001", "--compression", "zstd", "--mcap-compression", "none"],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    st = json.loads(r.stdout.strip().splitlines()[-1])
    assert st["messages"] == len(msgs) and st["converted"] == 3
    assert st["peak_held_bytes"] <= (64 + 2) << 20, st  # (the limit plus the message that crossed it), not 150 MB
    f = mcap_py.read(dst)
    assert len(f["messages"]) == len(msgs)
    for (ch, seq, lt, pt, data), (ch0, seq0, lt0, pt0, data0) in zip(f["messages"], msgs):
        assert (ch, seq, lt, pt) == (ch0, seq0, lt0, pt0)
        if ch0 == 1:
            assert data == data0
        else:
            assert data == reflib.ros_compress(np.frombuffer(data0, dtype=np.uint8), 0.001, 2).tobytes()
