# the fuzz campaign (random schemas, wide and very wide schemas, damaged streams, host-mirror streams): CLDN_FUZZ_EXTRA extra seeds per
# test; the commit it ran at is passed in (the GPU box has no .git): bash tools/runs/r5_fuzz.sh <git rev>
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
{ echo "# tests/test_gpu_fuzz.py with CLDN_FUZZ_EXTRA=${FUZZ_EXTRA:-30000} at commit ${1:-unknown}"; date -u;
  CLDN_FUZZ_EXTRA=${FUZZ_EXTRA:-30000} timeout 3300 python -m pytest tests/test_gpu_fuzz.py -q -m gpu -x 2>&1 | tail -4;
  echo "# the same tests with the chained launch of the point decoder forced (CLDN_HIP_NO_SPLIT_DECODE=1: small batches take the SPLIT launches by default), CLDN_FUZZ_EXTRA=3000";
  CLDN_HIP_NO_SPLIT_DECODE=1 CLDN_FUZZ_EXTRA=3000 timeout 1200 python -m pytest tests/test_gpu_fuzz.py -q -m gpu -x 2>&1 | tail -3; } | tee gpurun_out/r05_e_fuzz_campaign.txt
