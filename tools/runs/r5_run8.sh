export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_fused.py tests/test_device_lz4.py -x -q -m gpu 2>&1 | tail -6
timeout 600 python tools/lz4bench.py 2>&1 | grep -v amdgpu | grep x32
