export TMPDIR=/tmp
mkdir -p gpurun_out/r4
for rep in 1 2; do
for v in 0 1 3 5 7; do
cp cloudini_amd/lib/variants/libcloudini_hip_CLDN_FUSED_NT_$v.so cloudini_amd/lib/libcloudini_hip.so
echo -n "NT=$v "; timeout 300 python tools/finbench.py 2>&1 | grep -v amdgpu.ids
done; done | tee gpurun_out/r4/t23_nt.txt
