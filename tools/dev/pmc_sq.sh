#!/bin/bash
# one --pmc pass (SQ instruction mix and waits) over a command: tools/dev/pmc_sq.sh <tag> <command...> -> gpurun_out/<tag>_pmc_sq.txt
TAG=$1; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmcsq_$TAG
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS -d /tmp/pmcsq_$TAG -o trace -- "$@" > /tmp/pmcsq_$TAG.log 2>&1
db=$(find /tmp/pmcsq_$TAG -name "*.db" | head -1)
{ echo "# rocprofv3 --pmc SQ_* -- $*"; python $GRAFT_REPO_ROOT/tools/rocpd_summary.py $db --filter cldn; } > $OUT/${TAG}_pmc_sq.txt
