#!/bin/bash
# gpurun_out/ (what tools/runs/r4_evidence.sh left) -> profiles/r04_*
cd /root/repo
cp gpurun_out/prof_r04_a_kernel_trace.txt profiles/r04_a_kernel_trace.txt
cp gpurun_out/prof_r04_a_kernel_trace_bench.json profiles/r04_a_bench_under_rocprof.json
for k in fetch sq write; do cp gpurun_out/prof_r04_a_pmc_$k.txt profiles/r04_a_pmc_$k.txt; done
cp gpurun_out/r04_a_bench_default.json profiles/
for f in gpurun_out/r04_b_*.txt gpurun_out/r04_c_*.txt gpurun_out/r04_hbm_calib.txt; do b=$(basename $f); [ "$b" = "r04_b_transcode_c4.txt" ] && continue; cp $f profiles/$b; done
cp gpurun_out/prof_kt_r04_schema.txt profiles/r04_b_kernel_trace_schema.txt
python tools/make_traffic.py r04_a > /dev/null
