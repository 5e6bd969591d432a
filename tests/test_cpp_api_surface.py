"""The reference's public C++ headers as rebuilt under include/cloudini_lib/ must compile from a caller's point of
view and behave like the reference's in the parts that need no GPU (tests/cpp/api_surface.cpp)."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_public_headers_compile_and_behave(tmp_path):
    lib_dir = os.path.join(ROOT, "cloudini_amd", "lib")
    exe = str(tmp_path / "api_surface")
    subprocess.run(["g++", "-std=c++17", "-O1", "-Wall", "-I" + os.path.join(ROOT, "include"),
                    os.path.join(ROOT, "tests", "cpp", "api_surface.cpp"), os.path.join(lib_dir, "libcloudini_amd.so"),
                    os.path.join(lib_dir, "libcloudini_hip.so"), "-Wl,-rpath," + lib_dir, "-Wl,-rpath,/opt/rocm/lib",
                    "-o", exe], check=True)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "all checks passed" in r.stdout


def test_transcoder_pipeline_order_with_fake_stages(tmp_path):
    """tests/cpp/transcoder_order.cpp (no GPU): 1 / 3 / 5 stage threads of different speed keep the input order, a failing
    stage stops the run with a prefix written -- the logic a multi-GPU transcode (TranscodeOptions::devices) adds."""
    lib_dir = os.path.join(ROOT, "cloudini_amd", "lib")
    exe = str(tmp_path / "transcoder_order")
    subprocess.run(["g++", "-std=c++17", "-O1", "-Wall", "-pthread", "-I" + os.path.join(ROOT, "include"),
                    os.path.join(ROOT, "tests", "cpp", "transcoder_order.cpp"), os.path.join(lib_dir, "libcloudini_amd.so"),
                    os.path.join(lib_dir, "libcloudini_hip.so"), "-Wl,-rpath," + lib_dir, "-Wl,-rpath,/opt/rocm/lib",
                    "-o", exe], check=True)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "all checks passed" in r.stdout


import pytest  # noqa: E402


@pytest.mark.gpu
def test_cpp_caller_roundtrip_on_the_gpu(tmp_path):
    """tests/cpp/roundtrip_gpu.cpp: encode / decode / pre-filter through the C++ API only (no Python in the path)."""
    lib_dir = os.path.join(ROOT, "cloudini_amd", "lib")
    exe = str(tmp_path / "roundtrip_gpu")
    subprocess.run(["g++", "-std=c++17", "-O1", "-Wall", "-I" + os.path.join(ROOT, "include"),
                    os.path.join(ROOT, "tests", "cpp", "roundtrip_gpu.cpp"), os.path.join(lib_dir, "libcloudini_amd.so"),
                    os.path.join(lib_dir, "libcloudini_hip.so"), "-Wl,-rpath," + lib_dir, "-Wl,-rpath,/opt/rocm/lib",
                    "-o", exe], check=True)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "all checks passed" in r.stdout


@pytest.mark.gpu
def test_cpp_transcoder_sources_and_sinks(tmp_path):
    """tests/cpp/transcoder_sources.cpp: a sequential source with an order-checking sink and the concurrent directory
    source with a concurrent sink give the same messages (MessageSource::claim/fetch, MessageSink::concurrent)."""
    import sys
    sys.path.insert(0, ROOT)
    from cloudini_amd import synth
    src = tmp_path / "in"
    src.mkdir()
    for k in range(8):
        info, data = synth.velodyne_xyzir(3000 + 977 * k, seed=5 + k)
        synth.cdr_pointcloud2(info, data, stamp=(1700000000, k)).tofile(str(src / f"msg_{k:03d}.bin"))
    lib_dir = os.path.join(ROOT, "cloudini_amd", "lib")
    exe = str(tmp_path / "transcoder_sources")
    subprocess.run(["g++", "-std=c++17", "-O1", "-Wall", "-pthread", "-I" + os.path.join(ROOT, "include"),
                    os.path.join(ROOT, "tests", "cpp", "transcoder_sources.cpp"), os.path.join(lib_dir, "libcloudini_amd.so"),
                    os.path.join(lib_dir, "libcloudini_hip.so"), "-Wl,-rpath," + lib_dir, "-Wl,-rpath,/opt/rocm/lib",
                    "-o", exe], check=True)
    r = subprocess.run([exe, str(src)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "all checks passed" in r.stdout
