export TMPDIR=/tmp
mkdir -p gpurun_out/r4
timeout 1500 python -m pytest tests/test_gpu_decode.py tests/test_gpu_fuzz.py tests/test_gpu_encode.py -x -q -m gpu 2>&1 | tail -8 > gpurun_out/r4/t7_tests.txt
cat gpurun_out/r4/t7_tests.txt
timeout 600 python tools/schemabench.py > gpurun_out/r4/t7_schema.txt 2>&1
grep -v amdgpu.ids gpurun_out/r4/t7_schema.txt | grep -B1 decode | tail -30
bash tools/prof_any.sh r4_t7 python /root/repo/tools/schemabench.py > /dev/null 2>&1
cp gpurun_out/prof_kt_r4_t7.txt gpurun_out/r4/t7_schema_trace.txt
grep "k_decode\|k_mark\|k_locate\|k_sections\|k_build" gpurun_out/r4/t7_schema_trace.txt | head -30
