export TMPDIR=/tmp
timeout 2000 python -m pytest tests/test_gpu_encode.py tests/test_golden.py tests/test_gpu_fuzz.py tests/test_gpu_fused.py -x -q -m gpu 2>&1 | tail -3
SCHEMABENCH_ONLY=Gorilla timeout 300 python tools/schemabench.py 2>&1 | grep -v amdgpu | grep "Mpoints/s (" | cut -c1-170
bash tools/prof_any.sh r4_gor env SCHEMABENCH_ONLY=Gorilla python /root/repo/tools/schemabench.py 2>&1 | grep "k_gorilla\|k_encode_fused" | cut -c1-130
