export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_decode.py tests/test_gpu_fuzz.py tests/test_golden.py tests/test_gpu_fullsize.py -x -q -m gpu 2>&1 | tail -3
for c in c3 c4 c2; do timeout 300 python tools/decbench.py $c 2>&1 | grep -v amdgpu.ids | cut -c1-120; done
bash tools/prof_any.sh r4_c3dec python /root/repo/tools/decbench.py c3 2>&1 | grep "k_locate\|k_sections_cols_fast\|k_decode_points" | cut -c1-120
