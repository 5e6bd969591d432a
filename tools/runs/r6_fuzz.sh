# the round-6 fuzz campaign (random schemas, wide and very wide schemas, damaged streams, host-mirror streams) over a seed range no
# earlier campaign used: bash tools/runs/r6_fuzz.sh <git rev> [extra seeds per test, default 20000] [base, default 600000]
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
{ echo "# tests/test_gpu_fuzz.py with CLDN_FUZZ_EXTRA=${2:-20000} CLDN_FUZZ_BASE=${3:-600000} at commit ${1:-unknown}"; date -u;
  CLDN_FUZZ_EXTRA=${2:-20000} CLDN_FUZZ_BASE=${3:-600000} timeout 3300 python -m pytest tests/test_gpu_fuzz.py -q -m gpu -x 2>&1 | tail -4; } | tee gpurun_out/r06_e_fuzz_campaign.txt
