#!/bin/bash
# kernel trace of the default bench workload -> gpurun_out/prof_kt.txt
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof && rocprofv3 --kernel-trace -d /tmp/prof -o trace -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --cpu-baseline-seconds 0 > /tmp/prof.log 2>&1
tail -1 /tmp/prof.log
DB=$(find /tmp/prof -name "*.db" | head -1)
mkdir -p $GRAFT_REPO_ROOT/gpurun_out
python $GRAFT_REPO_ROOT/tools/rocpd_summary.py $DB --filter cldn | tee $GRAFT_REPO_ROOT/gpurun_out/prof_kt.txt
