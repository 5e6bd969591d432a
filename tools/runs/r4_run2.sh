export TMPDIR=/tmp
mkdir -p gpurun_out/r4
timeout 900 python -m pytest tests/test_gpu_decode.py tests/test_gpu_fullsize.py -x -q -m gpu 2>&1 | tail -5 > gpurun_out/r4/t2_tests.txt
cat gpurun_out/r4/t2_tests.txt
rm -f gpurun_out/r4/t2_decbench.txt
for k in tiles w8 w16 w8o6 w12o6; do
  echo "== $k" >> gpurun_out/r4/t2_decbench.txt
  CLDN_HIP_POINT_KERNEL=$k timeout 300 python tools/decbench.py c 2>&1 | grep -v amdgpu.ids | tail -6 >> gpurun_out/r4/t2_decbench.txt
done
cat gpurun_out/r4/t2_decbench.txt
CLDN_HIP_POINT_KERNEL=w16 bash tools/pmc_any.sh r4/t2_w16 python tools/decbench.py c2 > /dev/null 2>&1
grep -A12 "k_decode_points_w" gpurun_out/r4/t2_w16_pmc_sq.txt | head -40
