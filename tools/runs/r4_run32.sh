export TMPDIR=/tmp
export SCHEMABENCH_ONLY="1 us"
bash tools/prof_any.sh r4_dds1us python /root/repo/tools/schemabench.py 2>&1 | cut -c1-150 | head -40
