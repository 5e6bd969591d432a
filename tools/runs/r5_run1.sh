export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_gpu_encode.py tests/test_gpu_fused.py tests/test_golden.py tests/test_gpu_fullsize.py -x -q -m gpu 2>&1 | tail -4
bash tools/ab_env.sh - 2>&1 | grep -v amdgpu
