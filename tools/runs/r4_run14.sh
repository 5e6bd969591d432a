export TMPDIR=/tmp
mkdir -p gpurun_out/r4
timeout 600 python tools/groupserial.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r4/t14_groupserial.txt
