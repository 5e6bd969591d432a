export TMPDIR=/tmp
mkdir -p gpurun_out/r4
timeout 900 python -m pytest tests/test_gpu_decode.py -x -q -m gpu 2>&1 | tail -3
rm -f gpurun_out/r4/t4_decbench.txt
for k in tiles w8 w16 w8o6 w12o6; do
  echo "== $k" >> gpurun_out/r4/t4_decbench.txt
  CLDN_HIP_POINT_KERNEL=$k timeout 300 python tools/decbench.py c 2>&1 | grep -v amdgpu.ids | tail -6 >> gpurun_out/r4/t4_decbench.txt
done
cat gpurun_out/r4/t4_decbench.txt
for c in c2 c3; do
CLDN_HIP_POINT_KERNEL=w16 bash tools/pmc_any.sh r4/t4_w16_$c python /root/repo/tools/decbench.py $c > /dev/null 2>&1
grep -A8 "^cldn::k_decode_points_w" gpurun_out/r4/t4_w16_${c}_pmc_sq.txt | tail -9
done
