#!/bin/bash
# rocprofv3 passes over the default bench command (run on the GPU box through gpurun):
#   1 kernel trace + stats          2 --pmc FETCH_SIZE     3 --pmc WRITE_SIZE     4 --pmc SQ instruction mix
# Summaries land in gpurun_out/prof_<tag>_*.txt; copy the ones to keep into profiles/.
TAG=${1:-run}
BENCH="python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --cpu-baseline-seconds 0 --e2e-seconds 0 --transcode-messages 0 --config-legs 0"
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
run() {  # name, rocprofv3 args...
  local name=$1; shift
  rm -rf /tmp/prof_$name
  rocprofv3 "$@" -d /tmp/prof_$name -o trace -- $BENCH > /tmp/prof_$name.log 2>&1
  grep "^{\"metric\"" /tmp/prof_$name.log | tail -1 > $OUT/prof_${TAG}_${name}_bench.json
  local db=$(find /tmp/prof_$name -name "*.db" | head -1)
  { echo "# rocprofv3 $* -- $BENCH"; python $GRAFT_REPO_ROOT/tools/rocpd_summary.py $db --filter cldn; } > $OUT/prof_${TAG}_${name}.txt
}
run kernel_trace --kernel-trace --stats
run pmc_fetch --pmc FETCH_SIZE
run pmc_write --pmc WRITE_SIZE
run pmc_sq --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
cat $OUT/prof_${TAG}_kernel_trace.txt
grep -A3 "k_encode_fused\|k_encode_floatn\|k_compact\|k_section_palette32\|k_decode_points" $OUT/prof_${TAG}_pmc_fetch.txt $OUT/prof_${TAG}_pmc_write.txt | grep -v "^--"
