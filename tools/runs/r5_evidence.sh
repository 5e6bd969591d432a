# round-5 evidence: default-bench passes (r05_a), per-config / decode / schema / LZ4 / viz lines and traces (r05_b),
# decode and schema counter passes (r05_c), HBM calibration (plain kernels + the piece kernel's own shape), the default bench line.
# usage on the GPU box: bash tools/runs/r5_evidence.sh ; then (here) bash tools/runs/r5_copy_evidence.sh
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
bash tools/profile_round.sh r05_a > /dev/null 2>&1
bash tools/evidence_round.sh r05_b > /dev/null 2>&1
for c in c2 c3 c5; do bash tools/pmc_any.sh r05_c_dec_$c python /root/repo/tools/decbench.py $c > /dev/null 2>&1; done
bash tools/pmc_any.sh r05_c_schema python /root/repo/tools/schemabench.py > /dev/null 2>&1
bash tools/prof_any.sh r05_schema python /root/repo/tools/schemabench.py > /dev/null 2>&1
{ cloudini_amd/lib/hbm_calib 1; cloudini_amd/lib/hbm_calib 1 piece; } > gpurun_out/r05_hbm_calib.txt 2>&1
python bench.py > gpurun_out/r05_a_bench_default.json 2> gpurun_out/r05_a_bench_default.err
ls gpurun_out | grep r05 | head -80
