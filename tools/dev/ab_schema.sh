cd $GRAFT_REPO_ROOT
cp cloudini_amd/lib/libcloudini_hip.so /tmp/head.so
for rep in 1 2; do for v in sw0 sw1; do cp cloudini_amd/lib/variants/libcloudini_hip_$v.so cloudini_amd/lib/libcloudini_hip.so; for k in dds rgb ouster; do SCHEMABENCH_ONLY=$k python tools/schemabench.py 2>&1 | grep "decode" | sed "s/^/[$v $k] /" | cut -c1-60; done; done; done
cp /tmp/head.so cloudini_amd/lib/libcloudini_hip.so
