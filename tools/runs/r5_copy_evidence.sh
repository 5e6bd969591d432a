#!/bin/bash
# gpurun_out/ (what tools/runs/r5_evidence.sh left) -> profiles/r05_*
cd /root/repo
cp gpurun_out/prof_r05_a_kernel_trace.txt profiles/r05_a_kernel_trace.txt
cp gpurun_out/prof_r05_a_kernel_trace_bench.json profiles/r05_a_bench_under_rocprof.json
for k in fetch sq write; do cp gpurun_out/prof_r05_a_pmc_$k.txt profiles/r05_a_pmc_$k.txt; done
cp gpurun_out/r05_a_bench_default.json profiles/
for f in gpurun_out/r05_b_*.txt gpurun_out/r05_c_*.txt gpurun_out/r05_hbm_calib.txt gpurun_out/r05_e_fuzz_campaign.txt; do [ -f $f ] && cp $f profiles/$(basename $f); done
cp gpurun_out/prof_kt_r05_schema.txt profiles/r05_b_kernel_trace_schema.txt
python tools/make_traffic.py r05_a > /dev/null
ls profiles | grep r05
