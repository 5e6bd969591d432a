export TMPDIR=/tmp
mkdir -p gpurun_out/r4
for k in tiles w16; do
  for c in c2 c3; do
    CLDN_HIP_POINT_KERNEL=$k bash tools/pmc_any.sh r4/t3_${k}_${c} python /root/repo/tools/decbench.py $c > /dev/null 2>&1
    echo "== $k $c"; grep -A9 "k_decode_points" gpurun_out/r4/t3_${k}_${c}_pmc_sq.txt | grep -v "^--" | head -24
  done
done
