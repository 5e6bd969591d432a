export TMPDIR=/tmp
mkdir -p gpurun_out/r4
for rep in 1 2; do
for v in 4096u 2048u 1024u; do
cp cloudini_amd/lib/variants/libcloudini_hip_CLDN_PAL_SEED_$v.so cloudini_amd/lib/libcloudini_hip.so
if [ $rep = 1 ]; then timeout 600 python -m pytest tests/test_gpu_encode.py tests/test_golden.py -x -q -m gpu 2>&1 | tail -1; timeout 300 python tools/fintrace.py 2>&1 | grep "seed took\|pass 1 took\|build duration \|end at"; fi
echo -n "SEED=$v "; timeout 300 python tools/finbench.py 2>&1 | grep -v amdgpu.ids
done; done | tee gpurun_out/r4/t33_seed.txt
