export TMPDIR=/tmp
for rep in 1 2 3; do
for v in 0 1; do
cp cloudini_amd/lib/variants/libcloudini_hip_CLDN_WP_OPFIELD_$v.so cloudini_amd/lib/libcloudini_hip.so
echo -n "OPFIELD=$v "; SCHEMABENCH_ONLY=ouster timeout 300 python tools/schemabench.py 2>&1 | grep "decode" | cut -c1-60
done; done
