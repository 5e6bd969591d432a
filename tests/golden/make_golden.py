#!/usr/bin/env python3
"""Generate tests/golden/golden_v1.npz from the REAL reference (oracle/_ref, built from /root/reference).

Each entry stores the schema (the reference's own YAML header text + wire version), the input bytes and the
full stream the reference's PointcloudEncoder produced with CompressionOption::NONE (header + framed stage-1
chunks). LZ4/ZSTD streams are not stored: their bytes depend on the compressor library version.

Run in the build container only (needs /root/reference or a prebuilt oracle/_ref):
    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import cases  # noqa: E402
from cloudini_amd import synth  # noqa: E402
from cloudini_amd.schema import EncodingOptions  # noqa: E402
from oracle.binding import RefLib  # noqa: E402


def small_cases():
    out = []
    for name, info, data, _payload in cases.kat_vectors():
        out.append((name, info, data))
    out.append(("xyzi_struct_4133", *cases.xyzi_struct_4133()))
    out.append(("header_struct_v5", *cases.header_test_struct(1000, 5)))
    out.append(("header_struct_v4", *cases.header_test_struct(1000, 4)))
    out.append(("mixed_v5", *cases.mixed_schema(3000, 5)))
    out.append(("mixed_v4", *cases.mixed_schema(3000, 4)))
    out.append(("mixed_none", *cases.mixed_schema(1500, 5, EncodingOptions.NONE)))
    out.append(("mixed_lossless", *cases.mixed_schema_lossless(3000)))
    out.append(("five_floats", *cases.five_floats(2500)))
    out.append(("float_specials3", *cases.float_specials(n=3000, lanes=3)))
    out.append(("float_specials4", *cases.float_specials(n=3000, lanes=4, seed=4)))
    out.append(("c2_xyzi_5000", *synth.lidar_xyzi(5000)))
    out.append(("c3_depthcam_64x48", *synth.depthcam_xyzrgba(64, 48)))
    out.append(("c4_velodyne_6000", *synth.velodyne_xyzir(6000)))
    for name, info, data in cases.stride_variants():
        n = 2000
        out.append((name + "_2000", info.copy(width=n), data[: n * info.point_step]))
    for kind in ("grows_u16", "wide_u64", "two_values_i16"):
        info, data = cases.palette_stress(kind, n=6000)
        out.append((f"palette_{kind}_6000", info, data))
    for kind in ("long_and_short", "delta_runs_i64", "drle_then_noise"):
        info, data = cases.rle_stress(kind, n=7000)
        out.append((f"rle_{kind}_7000", info, data))
    out.append(("ouster_like_gorilla_5000", *cases.ouster_like(5000)))
    out.extend(reference_sample_slices())
    return out


def reference_sample_slices():
    """Slices of the reference's two sample files (data, not source): samples/dds_message.bin (CDR PointCloud2,
    64000 pts, XYZI f32 + ring u16 + f64 stamp -> Gorilla) and samples/lidar.pcd (binary PCD, XYZI f32)."""
    from cloudini_amd.schema import EncodingInfo, FieldType, PointField
    from cloudini_amd import api
    out = []
    samples = "/root/reference/cloudini_lib/samples"
    ref = RefLib()
    dds = np.fromfile(os.path.join(samples, "dds_message.bin"), dtype=np.uint8)
    yaml, off, size = ref.ros_describe(dds)
    info = api.parse_yaml_info(yaml, 5)
    for f in info.fields:
        if f.type == FieldType.FLOAT32:
            f.resolution = 0.001  # as cloudini_ros/src/conversion_utils.cpp:39 / test_ros_msg.cpp:135-138 set it
    n = 6000
    from cloudini_amd.schema import CompressionOption
    info = info.copy(width=n, height=1, compression_opt=CompressionOption.NONE, use_threads=False)
    out.append(("sample_dds_message_6000", info, dds[off: off + n * info.point_step].copy()))
    raw = open(os.path.join(samples, "lidar.pcd"), "rb").read()
    hdr_end = raw.index(b"DATA binary\n") + len(b"DATA binary\n")
    n = 8000
    fields = [PointField(c, 4 * k, FieldType.FLOAT32, 0.001) for k, c in enumerate(["x", "y", "z", "intensity"])]
    info = EncodingInfo(fields=fields, width=n, height=1, point_step=16, compression_opt=CompressionOption.NONE,
                        use_threads=False)
    out.append(("sample_lidar_pcd_8000", info, np.frombuffer(raw[hdr_end: hdr_end + n * 16], dtype=np.uint8).copy()))
    return out


def main():
    ref = RefLib()
    blob = {}
    names = []
    for name, info, data in small_cases():
        stream = ref.encode(info, data)
        header = ref.header(info)
        assert stream[: len(header)].tobytes() == header
        yaml = header[13:-1].decode()
        names.append(name)
        blob[name + "/yaml"] = np.frombuffer(yaml.encode(), dtype=np.uint8)
        blob[name + "/version"] = np.array([info.version], dtype=np.uint8)
        blob[name + "/input"] = np.ascontiguousarray(data)
        blob[name + "/stream"] = stream
        decoded, _ = ref.decode(stream, len(data), fill=0x5A)
        blob[name + "/decoded_fill5a"] = decoded
    blob["names"] = np.array(names)
    path = os.path.join(HERE, "golden_v1.npz")
    np.savez_compressed(path, **blob)
    print(f"{len(names)} cases -> {path} ({os.path.getsize(path)} bytes)")


if __name__ == "__main__":
    main()
