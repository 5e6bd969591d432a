export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_device_lz4.py tests/test_gpu_decode.py tests/test_gpu_fullsize.py tests/test_golden.py -x -q -m gpu 2>&1 | tail -4
timeout 600 python tools/lz4bench.py 2>&1 | grep -v amdgpu | grep x32
