export TMPDIR=/tmp
mkdir -p gpurun_out/r4
for rep in 1 2; do
for v in 4 2 1 0; do
cp cloudini_amd/lib/variants/libcloudini_hip_CLDN_WP_SLEEP_$v.so cloudini_amd/lib/libcloudini_hip.so
for c in c5 c2 s3 s4; do echo -n "SLEEP=$v "; timeout 300 python tools/decbench.py $c 2>&1 | grep -v amdgpu.ids | cut -c1-110; done
done; done | tee gpurun_out/r4/t35_sleep.txt
