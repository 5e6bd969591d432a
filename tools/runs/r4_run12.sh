export TMPDIR=/tmp
mkdir -p gpurun_out/r4
bash tools/prof_any.sh r4_t12 python /root/repo/tools/schemabench.py > /dev/null 2>&1
cp gpurun_out/prof_kt_r4_t12.txt gpurun_out/r4/t12_schema_trace.txt
grep "k_decode\|k_mark\|k_locate\|k_sections\|k_build\|k_walk" gpurun_out/r4/t12_schema_trace.txt | head -30
