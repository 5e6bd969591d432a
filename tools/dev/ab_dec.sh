#!/bin/bash
# same-box A/B of decode variants with tools/decbench.py: tools/dev/ab_dec.sh <case prefix|all> name1 name2 ... (head = the tree's library)
cd $GRAFT_REPO_ROOT
CASE=$1; shift
[ "$CASE" = all ] && CASE=""
cp cloudini_amd/lib/libcloudini_hip.so /tmp/libcloudini_hip_head.so
for rep in 1 2; do
for v in "$@"; do
  if [ $v = head ]; then cp /tmp/libcloudini_hip_head.so cloudini_amd/lib/libcloudini_hip.so; else cp cloudini_amd/lib/variants/libcloudini_hip_$v.so cloudini_amd/lib/libcloudini_hip.so; fi
  timeout 300 python tools/decbench.py $CASE 2>&1 | grep decode | sed "s/^/[$v] /"
done
done
cp /tmp/libcloudini_hip_head.so cloudini_amd/lib/libcloudini_hip.so
