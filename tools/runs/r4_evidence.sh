# round-4 evidence: default-bench passes (r04_a), per-config / decode / schema / LZ4 / viz lines and traces (r04_b),
# decode counter passes (r04_c), HBM calibration
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
bash tools/profile_round.sh r04_a > /dev/null 2>&1
bash tools/evidence_round.sh r04_b > /dev/null 2>&1
for c in c2 c3 c5; do bash tools/pmc_any.sh r04_c_dec_$c python /root/repo/tools/decbench.py $c > /dev/null 2>&1; done
bash tools/pmc_any.sh r04_c_schema python /root/repo/tools/schemabench.py > /dev/null 2>&1
bash tools/prof_any.sh r04_schema python /root/repo/tools/schemabench.py > /dev/null 2>&1
cloudini_amd/lib/hbm_calib 1 > gpurun_out/r04_hbm_calib.txt 2>&1
python bench.py > gpurun_out/r04_a_bench_default.json 2> gpurun_out/r04_a_bench_default.err
ls gpurun_out | grep r04 | head -60
