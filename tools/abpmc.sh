#!/bin/bash
# SQ instruction counters of k_encode_floatn for the baseline (A) and current library, same box
cd /tmp && export TMPDIR=/tmp
for lib in A cur; do
  if [ $lib = A ]; then export CLDN_HIP_LIB_OVERRIDE=$GRAFT_REPO_ROOT/cloudini_amd/lib/libcloudini_hip_A.so; else unset CLDN_HIP_LIB_OVERRIDE; fi
  rm -rf /tmp/p_$lib
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -d /tmp/p_$lib -o t -- python $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 2 --cpu-baseline-seconds 0 > /dev/null 2>&1
  echo "== $lib"; python $GRAFT_REPO_ROOT/tools/rocpd_summary.py $(find /tmp/p_$lib -name "*.db" | head -1) --filter "k_encode_floatn<256, 3, 2, 16384u, 4" | grep -E "SQ_|k_encode"
done
