export TMPDIR=/tmp; export FUZZ_EXTRA=30000
mkdir -p gpurun_out/r4
CLDN_FUZZ_EXTRA=${FUZZ_EXTRA:-5000} timeout 3000 python -m pytest tests/test_gpu_fuzz.py -q -m gpu -x 2>&1 | tail -4 | tee gpurun_out/r4/fuzz_campaign.txt
