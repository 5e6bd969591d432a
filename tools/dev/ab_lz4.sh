#!/bin/bash
# on the GPU box: tools/dev/ab_lz4.sh name1 name2 ... (variants of cloudini_amd/lib/variants; "head" = the built library)
cd $GRAFT_REPO_ROOT
cp cloudini_amd/lib/libcloudini_hip.so /tmp/libcloudini_hip_head.so
for rep in 1 2; do
for v in "$@"; do
  if [ $v = head ]; then cp /tmp/libcloudini_hip_head.so cloudini_amd/lib/libcloudini_hip.so; else cp cloudini_amd/lib/variants/libcloudini_hip_$v.so cloudini_amd/lib/libcloudini_hip.so; fi
  echo "[$v] $(timeout 300 python tools/dev/lz4_quick.py 2>&1 | tail -1)"
done
done
cp /tmp/libcloudini_hip_head.so cloudini_amd/lib/libcloudini_hip.so
