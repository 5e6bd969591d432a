export TMPDIR=/tmp
timeout 2000 python -m pytest tests/test_gpu_encode.py tests/test_golden.py tests/test_gpu_fuzz.py tests/test_gpu_fullsize.py tests/test_host_api.py -x -q -m gpu 2>&1 | tail -3
for e in X=1 CLDN_HIP_NO_FIXED_ENCODE=1; do echo -n "$e "; env $e SCHEMABENCH_ONLY=lossless timeout 300 python tools/schemabench.py 2>&1 | grep -v amdgpu | grep "Mpoints/s (" | cut -c1-160; done
