set -x
export TMPDIR=/tmp
mkdir -p gpurun_out/r4
timeout 900 python -m pytest tests/test_gpu_decode.py tests/test_gpu_fullsize.py -x -q -m gpu 2>&1 | tail -15 > gpurun_out/r4/t1_tests.txt
cat gpurun_out/r4/t1_tests.txt
for k in tiles w8 w16; do
  echo "== $k" >> gpurun_out/r4/t1_decbench.txt
  CLDN_HIP_POINT_KERNEL=$k timeout 300 python tools/decbench.py c 2>&1 | tail -6 >> gpurun_out/r4/t1_decbench.txt
done
cat gpurun_out/r4/t1_decbench.txt
timeout 300 cloudini_amd/lib/hbm_calib 1 > gpurun_out/r4/hbm_calib.txt 2>&1
tail -50 gpurun_out/r4/hbm_calib.txt
