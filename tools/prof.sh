#!/bin/bash
# kernel trace of one bench workload: tools/prof.sh <tag> <bench args...>  -> gpurun_out/prof_kt_<tag>.txt
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof && rocprofv3 --kernel-trace -d /tmp/prof -o trace -- python $GRAFT_REPO_ROOT/bench.py "$@" --steps 10 --warmup 2 --cpu-baseline-seconds 0 > /tmp/prof.log 2>&1
DB=$(find /tmp/prof -name "*.db" | head -1)
mkdir -p $GRAFT_REPO_ROOT/gpurun_out
{ echo "# bench.py $*"; python $GRAFT_REPO_ROOT/tools/rocpd_summary.py $DB --filter cldn; } | tee $GRAFT_REPO_ROOT/gpurun_out/prof_kt_$TAG.txt
