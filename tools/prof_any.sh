#!/bin/bash
# rocprofv3 kernel trace of an arbitrary command: tools/prof_any.sh <tag> <command...>; summary -> gpurun_out/prof_kt_<tag>.txt
TAG=$1; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_$TAG
rocprofv3 --kernel-trace --stats -d /tmp/prof_$TAG -o trace -- "$@" > $OUT/prof_kt_${TAG}_cmd.txt 2>&1
db=$(find /tmp/prof_$TAG -name "*.db" | head -1)
{ echo "# rocprofv3 --kernel-trace --stats -- $*"; python $GRAFT_REPO_ROOT/tools/rocpd_summary.py $db --filter cldn; } > $OUT/prof_kt_${TAG}.txt
cat $OUT/prof_kt_${TAG}.txt
