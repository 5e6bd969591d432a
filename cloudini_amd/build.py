"""Build the native libraries in-tree (hipcc cross-compiles gfx950 without a GPU).

  cloudini_amd/lib/libcloudini_hip.so   HIP kernels + the C ABI of include/cloudini_hip.h
  cloudini_amd/lib/libcloudini_amd.so   C++ host mirror of Cloudini::PointcloudEncoder/Decoder (+ C facade)

The .so files are git-ignored but travel to the GPU box with the gpurun snapshot.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
HIP_SO = os.path.join(LIBDIR, "libcloudini_hip.so")
HOST_SO = os.path.join(LIBDIR, "libcloudini_amd.so")

HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
HIP_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math",
             "-Wall", "-Wno-unused-function"]
LZ4_SO = "/usr/lib/x86_64-linux-gnu/liblz4.so.1"
ZSTD_SO = "/usr/lib/x86_64-linux-gnu/libzstd.so.1"


def _newer(target: str, sources) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in sources)


def _sources(folder: str, exts) -> list:
    out = []
    for base, _dirs, files in os.walk(folder):
        for f in files:
            if f.endswith(tuple(exts)):
                out.append(os.path.join(base, f))
    return sorted(out)


def build_hip(force: bool = False, verbose: bool = False) -> str:
    os.makedirs(LIBDIR, exist_ok=True)
    srcs = [os.path.join(CSRC, "stage1_kernels.hip"), os.path.join(CSRC, "viz_kernels.hip"),
            os.path.join(CSRC, "lz4_kernels.hip"), os.path.join(CSRC, "hip_abi.hip")]
    deps = _sources(CSRC, (".hip", ".h")) + _sources(os.path.join(ROOT, "include"), (".h",))
    deps = [d for d in deps if os.sep + "host" + os.sep not in d]
    if force or _newer(HIP_SO, deps):
        cmd = [HIPCC] + HIP_FLAGS + ["-shared", "-I" + os.path.join(ROOT, "include"), "-I" + CSRC] + srcs + \
              ["-o", HIP_SO]
        if verbose:
            print(" ".join(cmd))
        subprocess.run(cmd, check=True)
    return HIP_SO


def build_host(force: bool = False, verbose: bool = False) -> str:
    host_dir = os.path.join(CSRC, "host")
    srcs = _sources(host_dir, (".cpp",))
    if not srcs:
        return ""
    deps = srcs + _sources(host_dir, (".hpp", ".h")) + _sources(os.path.join(ROOT, "include"), (".h", ".hpp")) + \
        [HIP_SO]
    if force or _newer(HOST_SO, deps):
        # default visibility: the C++ API of include/cloudini_lib/*.hpp is what callers link against
        cmd = ["g++", "-std=c++17", "-O2", "-fPIC", "-shared", "-Wall",
               "-I" + os.path.join(ROOT, "include"), "-I" + host_dir] + srcs + \
              [HIP_SO, LZ4_SO, ZSTD_SO, "-lpthread", "-Wl,-rpath,$ORIGIN", "-o", HOST_SO]
        if verbose:
            print(" ".join(cmd))
        subprocess.run(cmd, check=True)
    return HOST_SO


TRANSCODE_BIN = os.path.join(LIBDIR, "cloudini_batch_transcode")


def build_tools(force: bool = False, verbose: bool = False) -> str:
    """Command-line batch transcoder (tools/cloudini_batch_transcode.cpp) next to the libraries."""
    src = os.path.join(ROOT, "tools", "cloudini_batch_transcode.cpp")
    if force or _newer(TRANSCODE_BIN, [src, HOST_SO]):
        cmd = ["g++", "-std=c++17", "-O2", "-Wall", "-I" + os.path.join(ROOT, "include"), src, HOST_SO, HIP_SO,
               "-lpthread", "-Wl,-rpath,$ORIGIN", "-o", TRANSCODE_BIN]
        if verbose:
            print(" ".join(cmd))
        subprocess.run(cmd, check=True)
    return TRANSCODE_BIN


def build_all(force: bool = False, verbose: bool = False) -> None:
    build_hip(force, verbose)
    build_host(force, verbose)
    build_tools(force, verbose)


if __name__ == "__main__":
    build_all(force="--force" in sys.argv, verbose=True)
