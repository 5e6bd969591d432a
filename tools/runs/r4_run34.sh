export TMPDIR=/tmp
mkdir -p gpurun_out/r4
for rep in 1 2 3; do
for v in 0 1 2 3 4; do
cp cloudini_amd/lib/variants/libcloudini_hip_CLDN_FIN_VAR_$v.so cloudini_amd/lib/libcloudini_hip.so
echo -n "VAR=$v "; timeout 300 python tools/finbench.py 2>&1 | grep -v amdgpu.ids
done; done | tee gpurun_out/r4/t34_finvar.txt
