#!/bin/bash
# one --pmc pass with an arbitrary counter set: tools/dev/pmc_set.sh <tag> "<counters>" <kernel filter> <command...> -> gpurun_out/<tag>.txt
TAG=$1; CTRS=$2; FILT=$3; shift 3
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmcset_$TAG
rocprofv3 --pmc $CTRS -d /tmp/pmcset_$TAG -o trace -- "$@" > /tmp/pmcset_$TAG.log 2>&1
db=$(find /tmp/pmcset_$TAG -name "*.db" | head -1)
{ echo "# rocprofv3 --pmc $CTRS -- $*"; python $GRAFT_REPO_ROOT/tools/rocpd_summary.py $db --filter $FILT; } > $OUT/$TAG.txt
