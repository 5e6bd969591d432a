export TMPDIR=/tmp
mkdir -p gpurun_out/r4
bash tools/prof_any.sh r4_s4 python /root/repo/tools/decbench.py s4 2>&1 | cut -c1-150 | tail -20
bash tools/prof_any.sh r4_s3 python /root/repo/tools/decbench.py s3 2>&1 | cut -c1-150 | tail -20
