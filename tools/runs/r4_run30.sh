export TMPDIR=/tmp
mkdir -p gpurun_out/r4
timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -5 | tee gpurun_out/r4/t30_tests.txt
CLDN_HIP_INTRA=0 timeout 900 python -m pytest tests/test_gpu_encode.py tests/test_golden.py tests/test_gpu_fused.py -x -q -m gpu 2>&1 | tail -3 | tee gpurun_out/r4/t30_tests_nointra.txt
CLDN_HIP_FINISH_COPY=1 timeout 900 python -m pytest tests/test_gpu_encode.py tests/test_golden.py -x -q -m gpu 2>&1 | tail -3 | tee gpurun_out/r4/t30_tests_copy1.txt
