#!/bin/bash
# the per-round evidence beside tools/profile_round.sh: one line per BASELINE config, decode / schema / LZ4 / viz timings
# and kernel traces of the configs the default bench does not run. usage (on the GPU box): tools/evidence_round.sh <tag>
TAG=${1:-run}
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
tools/allbench.sh > $OUT/${TAG}_allbench.txt 2>&1
python tools/decbench.py > $OUT/${TAG}_decbench.txt 2>&1
python tools/schemabench.py > $OUT/${TAG}_schema_bench.txt 2>&1
python tools/lz4bench.py > $OUT/${TAG}_lz4bench.txt 2>&1
python tools/vizbench.py > $OUT/${TAG}_viz_bench.txt 2>&1
# (tools/transcode_c4.py times ONE cold process of the command-line tool -- library load, page-locking of the batch buffers, first-call allocations included: 130-150 Mpoints/s; the warm in-process figure is the `transcode` object of the bench line)
tools/prof.sh c3 --workload c3 --clouds 16 --e2e-seconds 0 --transcode-messages 0 > /dev/null 2>&1; cp $OUT/prof_kt_c3.txt $OUT/${TAG}_kernel_trace_c3.txt
tools/prof.sh c4 --workload c4 --clouds 256 --e2e-seconds 0 --transcode-messages 0 > /dev/null 2>&1; cp $OUT/prof_kt_c4.txt $OUT/${TAG}_kernel_trace_c4.txt
tools/prof.sh c5 --workload c5 --clouds 1 --points 10000000 --e2e-seconds 0 --transcode-messages 0 > /dev/null 2>&1; cp $OUT/prof_kt_c5.txt $OUT/${TAG}_kernel_trace_c5.txt
tools/prof_any.sh dec python $GRAFT_REPO_ROOT/tools/decbench.py c > /dev/null 2>&1; cp $OUT/prof_kt_dec.txt $OUT/${TAG}_kernel_trace_decode.txt
tools/prof_any.sh viz python $GRAFT_REPO_ROOT/tools/vizbench.py > /dev/null 2>&1; cp $OUT/prof_kt_viz.txt $OUT/${TAG}_kernel_trace_viz.txt
tools/prof_any.sh lz4 python $GRAFT_REPO_ROOT/tools/lz4bench.py > /dev/null 2>&1; cp $OUT/prof_kt_lz4.txt $OUT/${TAG}_kernel_trace_lz4.txt
tail -n 12 $OUT/${TAG}_allbench.txt $OUT/${TAG}_decbench.txt $OUT/${TAG}_lz4bench.txt
