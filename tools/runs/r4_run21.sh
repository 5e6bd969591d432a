export TMPDIR=/tmp
mkdir -p gpurun_out/r4
timeout 900 python -m pytest tests/test_gpu_encode.py tests/test_golden.py tests/test_gpu_fused.py tests/test_gpu_fullsize.py -x -q -m gpu 2>&1 | tail -3
timeout 300 python tools/fintrace.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r4/t21_trace.txt
for rep in 1 2; do
for e in X=0 CLDN_HIP_FINISH_ABLATE=3; do
env $e timeout 300 python tools/finbench.py 2>&1 | grep -v amdgpu.ids
done; done | tee gpurun_out/r4/t21_fin.txt
