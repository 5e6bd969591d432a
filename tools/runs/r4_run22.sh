export TMPDIR=/tmp
mkdir -p gpurun_out/r4
timeout 900 python -m pytest tests/test_gpu_encode.py tests/test_golden.py tests/test_gpu_fused.py -x -q -m gpu 2>&1 | tail -2
for rep in 1 2 3; do
timeout 300 python tools/finbench.py 2>&1 | grep -v amdgpu.ids
done | tee gpurun_out/r4/t22_fin.txt
