#!/bin/bash
# same-box A/B of environment switches of the development flavour on tools/decbench.py: tools/dev/ab_dec_env.sh <case prefix> "VAR=val" "-" ...
cd $GRAFT_REPO_ROOT
CASE=$1; shift
cp cloudini_amd/lib/libcloudini_hip.so /tmp/libcloudini_hip_head.so
cp cloudini_amd/lib/variants/libcloudini_hip_dev.so cloudini_amd/lib/libcloudini_hip.so
for rep in 1 2; do
for envs in "$@"; do
  [ "$envs" = "-" ] && envs=""
  env $envs timeout 300 python tools/decbench.py $CASE 2>&1 | grep decode | sed "s/^/[$envs] /" | cut -c1-120
done
done
cp /tmp/libcloudini_hip_head.so cloudini_amd/lib/libcloudini_hip.so
