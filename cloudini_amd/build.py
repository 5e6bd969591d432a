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


def _compile_one(src: str, obj: str, verbose: bool) -> None:
    cmd = [HIPCC] + HIP_FLAGS + ["-c", "-MD", "-MF", obj + ".d", "-I" + os.path.join(ROOT, "include"), "-I" + CSRC, src, "-o", obj]
    if verbose:
        print(" ".join(cmd))
    subprocess.run(cmd, check=True)


def _depfile_sources(depfile: str):
    """Headers of this repository a translation unit included last time (make-style depfile of -MD), or None."""
    if not os.path.exists(depfile):
        return None
    text = open(depfile).read().replace("\\\n", " ")
    deps = [t for t in text.split()[1:] if t.startswith(ROOT) and os.path.exists(t)]
    return deps or None


def build_hip(force: bool = False, verbose: bool = False) -> str:
    """Every .hip translation unit is compiled to an object under build/ (side by side), then linked."""
    from concurrent.futures import ThreadPoolExecutor

    os.makedirs(LIBDIR, exist_ok=True)
    objdir = os.path.join(ROOT, "build", "hip")
    os.makedirs(objdir, exist_ok=True)
    names = ["stage1_kernels", "stage1_decode", "viz_kernels", "lz4_kernels", "hip_abi"]
    deps = _sources(CSRC, (".h",)) + _sources(os.path.join(ROOT, "include"), (".h",))
    deps = [d for d in deps if os.sep + "host" + os.sep not in d]
    jobs = []
    for n in names:
        src = os.path.join(CSRC, n + ".hip")
        obj = os.path.join(objdir, n + ".o")
        mine = _depfile_sources(obj + ".d") or (deps + [src])
        if force or _newer(obj, mine):
            jobs.append((src, obj))
    if jobs:
        with ThreadPoolExecutor(max_workers=len(jobs)) as ex:
            for f in [ex.submit(_compile_one, s, o, verbose) for s, o in jobs]:
                f.result()
    objs = [os.path.join(objdir, n + ".o") for n in names]
    if force or jobs or _newer(HIP_SO, objs):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", HIP_SO]
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


CALIB_BIN = os.path.join(LIBDIR, "hbm_calib")


def build_calib(force: bool = False, verbose: bool = False) -> str:
    """HBM calibrator (tools/hbm_calib.hip): what hand-written copy kernels reach on the box."""
    src = os.path.join(ROOT, "tools", "hbm_calib.hip")
    if force or _newer(CALIB_BIN, [src]):
        cmd = [HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", src, "-o", CALIB_BIN]
        if verbose:
            print(" ".join(cmd))
        subprocess.run(cmd, check=True)
    return CALIB_BIN


def build_all(force: bool = False, verbose: bool = False) -> None:
    build_hip(force, verbose)
    build_host(force, verbose)
    build_tools(force, verbose)
    build_calib(force, verbose)


if __name__ == "__main__":
    build_all(force="--force" in sys.argv, verbose=True)
