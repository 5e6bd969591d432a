export TMPDIR=/tmp
for rep in 1 2 3; do
for e in X=0 CLDN_HIP_FINISH_ORDER=3 CLDN_HIP_FINISH_ORDER=4 CLDN_HIP_FINISH_ORDER=2; do
env $e timeout 300 python tools/finbench.py 2>&1 | grep -v amdgpu.ids
done; done
