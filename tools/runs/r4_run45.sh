export TMPDIR=/tmp
timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -3
for e in X=1 CLDN_HIP_NO_FIXED_DECODE=1; do echo -n "$e "; env $e SCHEMABENCH_ONLY=lossless timeout 300 python tools/schemabench.py 2>&1 | grep -v amdgpu | grep "decode" | cut -c1-160; done
