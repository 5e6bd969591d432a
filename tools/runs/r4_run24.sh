export TMPDIR=/tmp
mkdir -p gpurun_out/r4
timeout 900 python -m pytest tests/test_gpu_encode.py tests/test_golden.py tests/test_gpu_fused.py tests/test_gpu_fullsize.py -x -q -m gpu 2>&1 | tail -2
for rep in 1 2; do
for v in 0 1 2 3; do
cp cloudini_amd/lib/variants/libcloudini_hip_CLDN_DEC_NT_$v.so cloudini_amd/lib/libcloudini_hip.so
for c in c2 c3 c5; do echo -n "DEC_NT=$v "; timeout 300 python tools/decbench.py $c 2>&1 | grep -v amdgpu.ids; done
done; done | tee gpurun_out/r4/t24_decnt.txt
cp cloudini_amd/lib/variants/libcloudini_hip_CLDN_DEC_NT_0.so cloudini_amd/lib/libcloudini_hip.so
bash tools/ab_env.sh - 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r4/t24_bench.txt
