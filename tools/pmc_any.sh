#!/bin/bash
# three --pmc passes (FETCH_SIZE, WRITE_SIZE, SQ instruction mix) over an arbitrary command, summaries per kernel:
#   tools/pmc_any.sh <tag> <command...>   -> gpurun_out/<tag>_pmc_fetch.txt, _pmc_write.txt, _pmc_sq.txt
# (counters only: never combined with a trace domain)
TAG=$1; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
run() {
  local name=$1; shift
  mkdir -p $(dirname /tmp/pmc_${TAG}_$name) $(dirname $OUT/${TAG}_x); rm -rf /tmp/pmc_${TAG}_$name
  rocprofv3 "$@" -d /tmp/pmc_${TAG}_$name -o trace -- "${CMD[@]}" > /tmp/pmc_${TAG}_$name.log 2>&1
  local db=$(find /tmp/pmc_${TAG}_$name -name "*.db" | head -1)
  { echo "# rocprofv3 $* -- ${CMD[*]}"; python $GRAFT_REPO_ROOT/tools/rocpd_summary.py $db --filter cldn; } > $OUT/${TAG}_${name}.txt
}
CMD=("$@")
run pmc_fetch --pmc FETCH_SIZE
run pmc_write --pmc WRITE_SIZE
run pmc_sq --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
