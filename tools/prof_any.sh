#!/bin/bash
# kernel trace of an arbitrary python command: tools/prof_any.sh <tag> <cmd...>
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof && rocprofv3 --kernel-trace -d /tmp/prof -o trace -- "$@" > /tmp/prof.log 2>&1
DB=$(find /tmp/prof -name "*.db" | head -1)
mkdir -p $GRAFT_REPO_ROOT/gpurun_out
python $GRAFT_REPO_ROOT/tools/rocpd_summary.py $DB --filter cldn | tee $GRAFT_REPO_ROOT/gpurun_out/prof_kt_$TAG.txt
