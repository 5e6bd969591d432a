export TMPDIR=/tmp
mkdir -p gpurun_out/r4
timeout 300 python tools/fintrace.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r4/t20_trace.txt
CLDN_HIP_FINISH_ORDER=1 timeout 300 python tools/fintrace.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r4/t20_trace_o1.txt
