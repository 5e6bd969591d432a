export TMPDIR=/tmp
mkdir -p gpurun_out/r4
timeout 900 python -m pytest tests/test_gpu_decode.py -x -q -m gpu 2>&1 | tail -3
rm -f gpurun_out/r4/t5_decbench.txt
for k in tiles w8 w16 w12o6; do
  echo "== $k" >> gpurun_out/r4/t5_decbench.txt
  CLDN_HIP_POINT_KERNEL=$k timeout 300 python tools/decbench.py c 2>&1 | grep -v amdgpu.ids | tail -6 >> gpurun_out/r4/t5_decbench.txt
done
cat gpurun_out/r4/t5_decbench.txt
