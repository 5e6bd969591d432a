# full GPU suite, then the evidence round: bash tools/runs/r5_final.sh <git rev>
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
{ echo "# python -m pytest tests -q -m gpu at commit ${1:-unknown}"; timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -3; } | tee gpurun_out/r05_gpu_suite.txt
bash tools/runs/r5_evidence.sh > /dev/null 2>&1
ls gpurun_out | grep -c r05
