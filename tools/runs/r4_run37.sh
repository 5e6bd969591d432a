export TMPDIR=/tmp
export SCHEMABENCH_ONLY="ouster"
bash tools/prof_any.sh r4_ouster python /root/repo/tools/schemabench.py 2>&1 | cut -c1-150 | head -30
