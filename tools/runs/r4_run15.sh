export TMPDIR=/tmp
mkdir -p gpurun_out/r4
for m in 1 2; do
CLDN_HIP_FINISH_COPY=$m timeout 900 python -m pytest tests/test_gpu_encode.py tests/test_golden.py tests/test_gpu_fused.py -x -q -m gpu 2>&1 | tail -3
done > gpurun_out/r4/t15_tests.txt 2>&1
cat gpurun_out/r4/t15_tests.txt
bash tools/ab_env.sh - CLDN_HIP_FINISH_COPY=1 CLDN_HIP_FINISH_COPY=2 CLDN_HIP_FINISH_ABLATE=1 CLDN_HIP_FINISH_ABLATE=2 CLDN_HIP_FINISH_ABLATE=3 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r4/t15_ab.txt
