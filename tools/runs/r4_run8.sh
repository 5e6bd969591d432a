export TMPDIR=/tmp
mkdir -p gpurun_out/r4
timeout 1500 python -m pytest tests/test_gpu_decode.py tests/test_gpu_fuzz.py tests/test_gpu_encode.py tests/test_golden.py -x -q -m gpu 2>&1 | tail -12 > gpurun_out/r4/t8_tests.txt
cat gpurun_out/r4/t8_tests.txt
timeout 600 python tools/schemabench.py > gpurun_out/r4/t8_schema.txt 2>&1
grep -v amdgpu.ids gpurun_out/r4/t8_schema.txt | grep -B1 decode | tail -30
