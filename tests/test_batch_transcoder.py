"""Batch transcoder (include/cloudini_amd/batch_transcoder.hpp): a directory of CDR PointCloud2 messages through ONE
batched GPU encode per schema run, stage 2 on the host pool, CDR wrapping -- every output message byte-identical to
cloudini_ros::convertPointCloud2ToCompressedCloud of the compiled reference (src/ros_msg_utils.cpp:167-213), which is
what the reference's converter loop (tools/src/mcap_converter.cpp:170-222) writes message by message."""
import json
import os
import subprocess

import numpy as np
import pytest

from cloudini_amd import api, synth
from cloudini_amd.schema import CompressionOption
from test_host_api import _cdr_pointcloud2

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _write_messages(folder, messages):
    os.makedirs(folder, exist_ok=True)
    for k, m in enumerate(messages):
        m.tofile(os.path.join(folder, f"msg_{k:05d}.bin"))


def _mixed_messages():
    msgs = []
    for k, n in enumerate([40000, 1, 70000, 0, 32768, 5000]):
        info, data = synth.lidar_xyzi(n, seed=10 + k)
        msgs.append(_cdr_pointcloud2(info, data, stamp=(1700000000 + k, 1000 * k)))
    info, data = synth.velodyne_xyzir(130048, seed=3)        # another schema in the middle of the bag
    msgs.append(_cdr_pointcloud2(info, data, frame_id="velodyne"))
    for k, n in enumerate([20000, 33000]):
        info, data = synth.lidar_xyzi(n, seed=50 + k)
        msgs.append(_cdr_pointcloud2(info, data, is_dense=False))
    info, data = synth.lidar_xyz(9000, seed=77)
    msgs.append(_cdr_pointcloud2(info, data))
    return msgs


@pytest.mark.parametrize("comp", [CompressionOption.ZSTD, CompressionOption.LZ4, CompressionOption.NONE])
def test_every_message_equals_the_reference_converter(tmp_path, reflib, comp):
    msgs = _mixed_messages()
    src, dst = str(tmp_path / "in"), str(tmp_path / "out")
    _write_messages(src, msgs)
    stats = api.transcode_directory(src, dst, resolution=0.001, compression_opt=int(comp), batch_messages=4)
    assert int(stats["messages"]) == len(msgs)
    assert int(stats["gpu_batches"]) >= 3  # 3 batches of 4 at least, more where the schema changes inside a batch
    for k, m in enumerate(msgs):
        got = np.fromfile(os.path.join(dst, f"msg_{k:05d}.bin"), dtype=np.uint8)
        want = reflib.ros_compress(m, 0.001, int(comp))
        assert got.size == want.size and np.array_equal(got, want), f"message {k}"
    # and the per-message path of the host mirror writes the same bytes
    assert np.array_equal(np.fromfile(os.path.join(dst, "msg_00002.bin"), dtype=np.uint8), api.ros_compress(msgs[2], 0.001, int(comp)))


def test_c4_shape_in_one_command(tmp_path, reflib):
    """BASELINE configs[3] end to end with the command-line tool: Velodyne-style XYZI+ring clouds of 130048 points, ZSTD
    second stage (64 messages here; tools/transcode_c4.py runs all 256)."""
    distinct = [synth.velodyne_xyzir(130048, seed=42 + k) for k in range(4)]
    msgs = [_cdr_pointcloud2(distinct[k % 4][0], distinct[k % 4][1], stamp=(1700000000, k)) for k in range(64)]
    src, dst = str(tmp_path / "in"), str(tmp_path / "out")
    _write_messages(src, msgs)
    exe = os.path.join(ROOT, "cloudini_amd", "lib", "cloudini_batch_transcode")
    r = subprocess.run([exe, src, dst, "--resolution", "0.001", "--compression", "zstd", "--batch", "32"],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr
    st = json.loads(r.stdout.strip().splitlines()[-1])
    assert st["messages"] == 64 and st["gpu_batches"] == 2 and st["points"] == 64 * 130048
    for k in (0, 1, 2, 3, 37, 63):
        got = np.fromfile(os.path.join(dst, f"msg_{k:05d}.bin"), dtype=np.uint8)
        assert np.array_equal(got, reflib.ros_compress(msgs[k], 0.001, int(CompressionOption.ZSTD))), k


def test_viz_prefilter_in_the_batch(tmp_path):
    info, data = synth.lidar_xyzi(60000, seed=5)
    pts = data.reshape(-1, 16).copy()
    pts[::7, 0:4] = np.frombuffer(np.float32(np.nan).tobytes(), dtype=np.uint8)   # NaN x in every 7th point
    msgs = [_cdr_pointcloud2(info, pts.reshape(-1))] * 3
    src, dst = str(tmp_path / "in"), str(tmp_path / "out")
    _write_messages(src, msgs)
    stats = api.transcode_directory(src, dst, resolution=0.01, compression_opt=int(CompressionOption.ZSTD), viz_lossy=True)
    assert int(stats["messages"]) == 3 and 0 < stats["points"] < 3 * (60000 - 60000 // 7)
    out = np.fromfile(os.path.join(dst, "msg_00000.bin"), dtype=np.uint8)
    back = api.ros_decompress(out, msgs[0].size)
    assert back.size < msgs[0].size  # fewer points come back than went in


@pytest.mark.parametrize("comp", [CompressionOption.ZSTD, CompressionOption.LZ4, CompressionOption.NONE])
def test_the_way_back_equals_the_reference_converter(tmp_path, reflib, comp):
    """CompressedPointCloud2 messages (written by the reference) through the decode direction of the transcoder: stage 2
    undone on the pool, one batched GPU decode per schema run, CDR wrapping -- every output message byte-identical to
    cloudini_ros::convertCompressedCloudToPointCloud2 of the reference (src/ros_msg_utils.cpp:135-165), which is what
    McapConverter::decodePointClouds (tools/src/mcap_converter.cpp:240-300) writes message by message."""
    msgs = _mixed_messages()
    packed = [reflib.ros_compress(m, 0.001, int(comp)) for m in msgs]
    src, dst = str(tmp_path / "in"), str(tmp_path / "out")
    _write_messages(src, packed)
    stats = api.decode_directory(src, dst, batch_messages=4)
    assert int(stats["messages"]) == len(msgs)
    assert int(stats["points"]) == 40000 + 1 + 70000 + 0 + 32768 + 5000 + 130048 + 20000 + 33000 + 9000
    for k, m in enumerate(packed):
        got = np.fromfile(os.path.join(dst, f"msg_{k:05d}.bin"), dtype=np.uint8)
        want = reflib.ros_decompress(m, msgs[k].size + 4096)
        assert got.size == want.size and np.array_equal(got, want), f"message {k}"
    # the per-message path of the host mirror writes the same bytes
    assert np.array_equal(np.fromfile(os.path.join(dst, "msg_00002.bin"), dtype=np.uint8), api.ros_decompress(packed[2], msgs[2].size + 4096))


def test_round_trip_through_both_directions_with_the_tool(tmp_path):
    info, data = synth.velodyne_xyzir(130048, seed=8)
    msgs = [_cdr_pointcloud2(info, data, stamp=(1700000000, k)) for k in range(6)]
    a, bdir, c = str(tmp_path / "a"), str(tmp_path / "b"), str(tmp_path / "c")
    _write_messages(a, msgs)
    exe = os.path.join(ROOT, "cloudini_amd", "lib", "cloudini_batch_transcode")
    r = subprocess.run([exe, a, bdir, "--resolution", "0.001", "--compression", "zstd", "--batch", "4"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr
    r = subprocess.run([exe, bdir, c, "--decode", "--batch", "4"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr
    st = json.loads(r.stdout.strip().splitlines()[-1])
    assert st["messages"] == 6 and st["points"] == 6 * 130048
    back = np.fromfile(os.path.join(c, "msg_00003.bin"), dtype=np.uint8)
    assert back.size == msgs[3].size                      # same CDR layout, points within half a tick
    n = 130048 * info.point_step
    assert np.array_equal(back[: msgs[3].size - n - 1], msgs[3][: msgs[3].size - n - 1])  # header part unchanged


def test_two_gpu_stages_keep_order_and_bytes(tmp_path, reflib):
    """TranscodeOptions::devices: one GPU stage per entry, shared reader / stage-2 pool / ordered writer. Two stages on
    device 0 (what a 1-GPU box can run) must write the files the single-stage run writes, in the same order; the
    command-line tool takes the list as --devices."""
    msgs = _mixed_messages() * 3
    src, one, two, three = str(tmp_path / "in"), str(tmp_path / "one"), str(tmp_path / "two"), str(tmp_path / "three")
    _write_messages(src, msgs)
    api.transcode_directory(src, one, resolution=0.001, compression_opt=int(CompressionOption.ZSTD), batch_messages=3)
    st = api.transcode_directory(src, two, resolution=0.001, compression_opt=int(CompressionOption.ZSTD), batch_messages=3,
                                 devices=[0, 0])
    assert int(st["messages"]) == len(msgs)
    exe = os.path.join(ROOT, "cloudini_amd", "lib", "cloudini_batch_transcode")
    r = subprocess.run([exe, src, three, "--resolution", "0.001", "--compression", "zstd", "--batch", "3", "--devices", "0,0,0"],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr
    assert json.loads(r.stdout.strip().splitlines()[-1])["gpu_stages"] == 3
    for k, m in enumerate(msgs):
        a = np.fromfile(os.path.join(one, f"msg_{k:05d}.bin"), dtype=np.uint8)
        for other in (two, three):
            b = np.fromfile(os.path.join(other, f"msg_{k:05d}.bin"), dtype=np.uint8)
            assert a.size == b.size and np.array_equal(a, b), (other, k)
    assert np.array_equal(np.fromfile(os.path.join(two, "msg_00006.bin"), dtype=np.uint8),
                          reflib.ros_compress(msgs[6], 0.001, int(CompressionOption.ZSTD)))
    # the way back over two stages
    back1, back2 = str(tmp_path / "b1"), str(tmp_path / "b2")
    api.decode_directory(one, back1, batch_messages=3)
    api.decode_directory(one, back2, batch_messages=3, devices=[0, 0])
    for k in range(len(msgs)):
        assert np.array_equal(np.fromfile(os.path.join(back1, f"msg_{k:05d}.bin"), dtype=np.uint8),
                              np.fromfile(os.path.join(back2, f"msg_{k:05d}.bin"), dtype=np.uint8)), k
    with pytest.raises(RuntimeError):
        api.transcode_directory(src, str(tmp_path / "bad"), devices=[99])


@pytest.mark.skipif("__import__('torch').cuda.device_count() < 2")
def test_two_real_gpus_write_the_single_gpu_bytes(tmp_path):
    msgs = _mixed_messages() * 2
    src, one, two = str(tmp_path / "in"), str(tmp_path / "one"), str(tmp_path / "two")
    _write_messages(src, msgs)
    api.transcode_directory(src, one, batch_messages=2)
    api.transcode_directory(src, two, batch_messages=2, devices=[0, 1])
    for k in range(len(msgs)):
        assert np.array_equal(np.fromfile(os.path.join(one, f"msg_{k:05d}.bin"), dtype=np.uint8),
                              np.fromfile(os.path.join(two, f"msg_{k:05d}.bin"), dtype=np.uint8)), k


def test_truncated_and_degenerate_messages_fail_like_the_reference(tmp_path, reflib):
    """The reference's order, message by message (src/ros_msg_utils.cpp:178-190, src/cloudini.cpp:525-531): an empty cloud
    is an empty message whatever its schema says; a data blob that is not a multiple of point_step is an error, never a
    silently shortened cloud."""
    info, data = synth.lidar_xyzi(1000, seed=1)
    good = _cdr_pointcloud2(info, data)
    cut = _cdr_pointcloud2(info, data[:-5])            # 995 whole points + 11 bytes
    tiny = _cdr_pointcloud2(info, data[:7])            # less than one point
    for bad in (cut, tiny):
        with pytest.raises(RuntimeError, match="not a multiple of point_step"):
            reflib.ros_compress(bad, 0.001, int(CompressionOption.ZSTD))
        src = str(tmp_path / f"in{bad.size}")
        _write_messages(src, [good, bad, good])
        with pytest.raises(RuntimeError, match="not a multiple of point_step"):
            api.transcode_directory(src, str(tmp_path / f"out{bad.size}"), batch_messages=4)
    # an empty cloud with point_step 0 in its schema: an empty message in the reference, and here
    empty0 = _cdr_pointcloud2(info.copy(point_step=0, width=0), data[:0])
    src, dst = str(tmp_path / "in_e"), str(tmp_path / "out_e")
    _write_messages(src, [good, empty0, good])
    st = api.transcode_directory(src, dst, batch_messages=4)
    assert int(st["messages"]) == 3
    assert np.array_equal(np.fromfile(os.path.join(dst, "msg_00001.bin"), dtype=np.uint8),
                          reflib.ros_compress(empty0, 0.001, int(CompressionOption.ZSTD)))


def test_wrapping_geometry_is_refused_on_the_way_back(tmp_path):
    """width * height * point_step of a CompressedPointCloud2 must not wrap 64 bits into a small (or zero) cloud size."""
    info, data = synth.lidar_xyzi(100, seed=1)
    packed = api.ros_compress(_cdr_pointcloud2(info, data), 0.001, int(CompressionOption.NONE))
    msg = bytearray(packed.tobytes())
    # height and width sit behind stamp (8) + frame_id string; patch both to 2^31 -> 2^62 points * 16 bytes wraps to 0
    at = 4 + 8 + 4 + len(b"lidar_top\0")
    at += (-(at - 4)) % 4
    assert int.from_bytes(msg[at:at + 4], "little") == info.height and int.from_bytes(msg[at + 4:at + 8], "little") == info.width
    msg[at:at + 4] = (1 << 31).to_bytes(4, "little")
    msg[at + 4:at + 8] = (1 << 31).to_bytes(4, "little")
    src = str(tmp_path / "in")
    _write_messages(src, [np.frombuffer(bytes(msg), dtype=np.uint8)])
    with pytest.raises(RuntimeError):
        api.decode_directory(src, str(tmp_path / "out"))


def _swap_stream(msg: np.ndarray, new_stream: bytes) -> np.ndarray:
    """A CompressedPointCloud2 CDR message with its compressed_data replaced: [u32 length][bytes] is followed by is_dense
    (1 byte) and the format string ([pad to 4][u32 length]["cloudini\0"]), src/ros_msg_utils.cpp:167-213."""
    raw = msg.tobytes()
    at = raw.find(b"CLOUDINI_V")
    assert at >= 8
    old_len = int.from_bytes(raw[at - 4:at], "little")
    tail = raw[at + old_len:]
    is_dense = tail[:1]
    k = tail.find(b"cloudini")
    fmt = tail[k - 4:]                                   # [u32 9]["cloudini\0"]
    head = raw[:at - 4] + len(new_stream).to_bytes(4, "little") + new_stream + is_dense
    pad = (-(len(head) - 4)) % 4                         # CDR aligns relative to the end of the 4-byte encapsulation header
    return np.frombuffer(head + b"\0" * pad + fmt, dtype=np.uint8)


def test_wire_version_2_messages_in_a_bag_decode_one_by_one(tmp_path, reflib):
    """A CompressedPointCloud2 whose stream has wire version 2 (no chunk framing: one unframed payload,
    src/cloudini.cpp:665-667) between version-5 messages: the batched way back must not read its first bytes as a chunk
    size -- it takes the single-message path, and every output equals the reference's converter."""
    from test_host_api import _stage2
    msgs, packed = [], []
    for k, n in enumerate([9000, 20000, 7000]):
        info, data = synth.lidar_xyzi(n, seed=70 + k)
        m = _cdr_pointcloud2(info, data, stamp=(1700000100 + k, 5 * k))
        msgs.append(m)
        packed.append(reflib.ros_compress(m, 0.001, int(CompressionOption.NONE)))
    # message 1 becomes a version-2 stream of the same cloud: header "CLOUDINI_V02" + the stage-1 payload without its prefix
    info, data = synth.lidar_xyzi(20000, seed=71)
    info3 = info.copy(version=3, compression_opt=CompressionOption.NONE)
    framed = reflib.encode_stage1(info3, data)
    assert int.from_bytes(framed[:4].tobytes(), "little") + 4 == framed.size  # one chunk
    for comp in (CompressionOption.NONE, CompressionOption.ZSTD):
        info2 = info3.copy(version=2, compression_opt=comp, width=20000, height=1)
        v2 = reflib.header(info2) + _stage2(comp, framed[4:].tobytes())
        bag = [packed[0], _swap_stream(packed[1], v2), packed[2]]
        want1 = reflib.ros_decompress(bag[1], msgs[1].size + 4096)      # the reference reads the message
        src, dst = str(tmp_path / f"in{int(comp)}"), str(tmp_path / f"out{int(comp)}")
        _write_messages(src, bag)
        stats = api.decode_directory(src, dst, batch_messages=4)
        assert int(stats["messages"]) == 3
        for k in range(3):
            got = np.fromfile(os.path.join(dst, f"msg_{k:05d}.bin"), dtype=np.uint8)
            want = want1 if k == 1 else reflib.ros_decompress(bag[k], msgs[k].size + 4096)
            assert got.size == want.size and np.array_equal(got, want), (int(comp), k)
