export TMPDIR=/tmp
mkdir -p gpurun_out/r4
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -8 > gpurun_out/r4/t6_tests.txt
cat gpurun_out/r4/t6_tests.txt
timeout 600 python tools/schemabench.py > gpurun_out/r4/t6_schema.txt 2>&1
CLDN_HIP_NO_STREAM_KERNEL=1 timeout 600 python tools/schemabench.py > gpurun_out/r4/t6_schema_old.txt 2>&1
grep -v amdgpu.ids gpurun_out/r4/t6_schema.txt | tail -30
echo ---- old
grep -v amdgpu.ids gpurun_out/r4/t6_schema_old.txt | grep -i "decode" | tail -12
