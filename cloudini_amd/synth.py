"""Deterministic synthetic point clouds for BASELINE.json's five configurations.

The reference ships no generator for these (its benchmarks read private rosbags); the shapes follow the
LiDAR / depth-camera layouts of its fixtures (cloudini_lib/samples/lidar.pcd: XYZI float32;
samples/dds_message.bin: XYZI float32 + ring u16 + f64 stamp) and of cloudini_ros/src/conversion_utils.cpp.
All generators are pure numpy, seeded, and return (EncodingInfo, uint8 array of n*point_step bytes).
"""
from __future__ import annotations

import numpy as np

from .schema import CompressionOption, EncodingInfo, EncodingOptions, FieldType, PointField


def _rng(seed: int) -> np.random.RandomState:
    return np.random.RandomState(seed)  # MT19937, as std::mt19937(seed)


def _lidar_xyz(n: int, rings: int, rs: np.random.RandomState):
    i = np.arange(n, dtype=np.int64)
    ring = (i % rings).astype(np.float32)
    col = (i // rings).astype(np.float32)
    cols = np.float32(max(1, -(-n // rings)))
    az = np.float32(2.0 * np.pi) * col / cols
    el = np.deg2rad(np.float32(-25.0) + np.float32(40.0) * ring / np.float32(max(1, rings - 1))).astype(np.float32)
    r = (np.float32(20.0) + np.float32(10.0) * np.sin(np.float32(3.0) * az)
         + np.float32(5.0) * np.cos(np.float32(7.0) * az + np.float32(0.1) * ring)
         + rs.normal(0.0, 0.002, size=n).astype(np.float32)).astype(np.float32)
    x = (r * np.cos(el) * np.cos(az)).astype(np.float32)
    y = (r * np.cos(el) * np.sin(az)).astype(np.float32)
    z = (r * np.sin(el)).astype(np.float32)
    return x, y, z, (i % rings), (i // rings)


def xyz_info(n: int, res: float = 0.001, compression=CompressionOption.NONE) -> EncodingInfo:
    return EncodingInfo(
        fields=[PointField("x", 0, FieldType.FLOAT32, res), PointField("y", 4, FieldType.FLOAT32, res),
                PointField("z", 8, FieldType.FLOAT32, res)],
        width=n, height=1, point_step=12, encoding_opt=EncodingOptions.LOSSY, compression_opt=compression)


def lidar_xyz(n: int, seed: int = 42, res: float = 0.001, rings: int = 64):
    """C1 / C5: dense XYZ float32, point_step 12."""
    x, y, z, _, _ = _lidar_xyz(n, rings, _rng(seed))
    pts = np.empty((n, 3), dtype=np.float32)
    pts[:, 0], pts[:, 1], pts[:, 2] = x, y, z
    return xyz_info(n, res), pts.view(np.uint8).reshape(-1)


def uniform_xyz(n: int, seed: int = 42, res: float = 0.001, lo: float = -50.0, hi: float = 50.0):
    """Worst case: spatially incoherent points (about 2.8 bytes per varint)."""
    pts = _rng(seed).uniform(lo, hi, size=(n, 3)).astype(np.float32)
    return xyz_info(n, res), pts.view(np.uint8).reshape(-1)


def xyzi_info(n: int, res: float = 0.001, compression=CompressionOption.NONE) -> EncodingInfo:
    return EncodingInfo(
        fields=[PointField("x", 0, FieldType.FLOAT32, res), PointField("y", 4, FieldType.FLOAT32, res),
                PointField("z", 8, FieldType.FLOAT32, res), PointField("intensity", 12, FieldType.UINT16, None)],
        width=n, height=1, point_step=16, encoding_opt=EncodingOptions.LOSSY, compression_opt=compression)


def lidar_xyzi(n: int, seed: int = 42, res: float = 0.001, rings: int = 64):
    """C2 (the headline config): XYZ float32 + uint16 intensity, point_step 16. Intensity has 256 levels
    (-> V5 Palette section)."""
    rs = _rng(seed)
    x, y, z, _, _ = _lidar_xyz(n, rings, rs)
    dt = np.dtype({"names": ["x", "y", "z", "i", "pad"], "formats": ["<f4", "<f4", "<f4", "<u2", "<u2"],
                   "offsets": [0, 4, 8, 12, 14], "itemsize": 16})
    pts = np.zeros(n, dtype=dt)
    pts["x"], pts["y"], pts["z"] = x, y, z
    pts["i"] = (rs.randint(0, 256, size=n).astype(np.uint16) * 16)
    return xyzi_info(n, res), pts.view(np.uint8).reshape(-1)


def xyzrgba_info(w: int, h: int, res: float = 0.0001, compression=CompressionOption.NONE) -> EncodingInfo:
    return EncodingInfo(
        fields=[PointField("x", 0, FieldType.FLOAT32, res), PointField("y", 4, FieldType.FLOAT32, res),
                PointField("z", 8, FieldType.FLOAT32, res), PointField("rgba", 16, FieldType.UINT32, None)],
        width=w, height=h, point_step=32, encoding_opt=EncodingOptions.LOSSY, compression_opt=compression)


def depthcam_xyzrgba(w: int = 1280, h: int = 800, seed: int = 42, res: float = 0.0001, nan_every: int = 20):
    """C3: organised Realsense-style cloud, 32-byte stride with padding, 5 % NaN pixels."""
    rs = _rng(seed)
    n = w * h
    v, u = np.divmod(np.arange(n, dtype=np.int64), w)
    uf, vf = u.astype(np.float32), v.astype(np.float32)
    z = (np.float32(1.5) + np.float32(0.5) * np.sin(uf / np.float32(97.0)) * np.cos(vf / np.float32(61.0))
         + rs.normal(0.0, 5e-4, size=n).astype(np.float32)).astype(np.float32)
    x = ((uf - np.float32(w / 2)) * z / np.float32(640.0)).astype(np.float32)
    y = ((vf - np.float32(h / 2)) * z / np.float32(640.0)).astype(np.float32)
    bad = rs.randint(0, nan_every, size=n) == 0
    x[bad] = np.nan
    y[bad] = np.nan
    z[bad] = np.nan
    rgba = (np.uint32(0xFF000000) + (((u * 4) & 0xFF).astype(np.uint32) << 16)
            + ((v & 0xFF).astype(np.uint32) << 8) + rs.randint(0, 16, size=n).astype(np.uint32))
    dt = np.dtype({"names": ["x", "y", "z", "rgba"], "formats": ["<f4", "<f4", "<f4", "<u4"],
                   "offsets": [0, 4, 8, 16], "itemsize": 32})
    pts = np.zeros(n, dtype=dt)
    pts["x"], pts["y"], pts["z"], pts["rgba"] = x, y, z, rgba
    return xyzrgba_info(w, h, res), pts.view(np.uint8).reshape(-1)


def velodyne_info(n: int, res: float = 0.001, compression=CompressionOption.NONE) -> EncodingInfo:
    return EncodingInfo(
        fields=[PointField("x", 0, FieldType.FLOAT32, res), PointField("y", 4, FieldType.FLOAT32, res),
                PointField("z", 8, FieldType.FLOAT32, res), PointField("intensity", 12, FieldType.FLOAT32, res),
                PointField("ring", 16, FieldType.UINT16, None)],
        width=n, height=1, point_step=18, encoding_opt=EncodingOptions.LOSSY, compression_opt=compression)


def velodyne_xyzir(n: int = 130048, seed: int = 42, res: float = 0.001, rings: int = 128):
    """C4: XYZI float32 (4 fused lossy floats) + ring uint16, packed 18-byte points (unaligned)."""
    rs = _rng(seed)
    x, y, z, ring, _ = _lidar_xyz(n, rings, rs)
    dt = np.dtype({"names": ["x", "y", "z", "i", "ring"], "formats": ["<f4", "<f4", "<f4", "<f4", "<u2"],
                   "offsets": [0, 4, 8, 12, 16], "itemsize": 18})
    pts = np.zeros(n, dtype=dt)
    pts["x"], pts["y"], pts["z"] = x, y, z
    pts["i"] = rs.randint(0, 256, size=n).astype(np.float32)
    pts["ring"] = ring.astype(np.uint16)
    return velodyne_info(n, res), pts.view(np.uint8).reshape(-1)


def cdr_pointcloud2(info, data, frame_id="lidar_top", stamp=(1700000000, 123456789), is_dense=True):
    """Serialise a sensor_msgs/PointCloud2 as little-endian CDR (what a DDS reader hands to the converter)."""
    out = bytearray(b"\x00\x01\x00\x00")

    def align(n):
        while (len(out) - 4) % n:
            out.append(0)

    def u32(v):
        align(4)
        out.extend(int(v).to_bytes(4, "little"))

    def string(s):
        b = s.encode() + b"\0"
        u32(len(b))
        out.extend(b)

    align(4)
    out.extend(int(stamp[0]).to_bytes(4, "little", signed=True))
    u32(stamp[1])
    string(frame_id)
    u32(info.height)
    u32(info.width)
    u32(len(info.fields))
    for f in info.fields:
        string(f.name)
        u32(f.offset)
        out.append(int(f.type))
        u32(1)
    out.append(0)  # is_bigendian
    u32(info.point_step)
    u32(info.point_step * info.width)
    u32(len(data))
    out.extend(bytes(data))
    out.append(1 if is_dense else 0)
    return np.frombuffer(bytes(out), dtype=np.uint8)
