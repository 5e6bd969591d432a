export TMPDIR=/tmp
mkdir -p gpurun_out/r4
timeout 900 python -m pytest tests/test_device_lz4.py -x -q 2>&1 | tail -3
timeout 600 python tools/lz4bench.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r4/t27_lz4.txt
bash tools/prof_any.sh r4_lz4 python /root/repo/tools/lz4bench.py 2>&1 | grep "k_lz4" | cut -c1-150
