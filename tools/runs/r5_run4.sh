export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_device_lz4.py tests/test_host_api.py tests/test_gpu_decode.py -x -q -m gpu 2>&1 | tail -4
timeout 600 python bench.py > gpurun_out/r05_bench1.json 2> gpurun_out/r05_bench1.err; tail -3 gpurun_out/r05_bench1.err
timeout 600 python tools/lz4bench.py 2>&1 | grep -v amdgpu | tee gpurun_out/r05_lz4bench.txt
