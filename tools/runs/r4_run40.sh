export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_decode.py tests/test_gpu_fuzz.py tests/test_golden.py tests/test_gpu_fullsize.py -x -q -m gpu 2>&1 | tail -2
for rep in 1 2 3; do
for e in X=1 CLDN_HIP_NO_STORE_MODES=1; do
for c in c2 c5; do echo -n "$e "; env $e timeout 300 python tools/decbench.py $c 2>&1 | grep -v amdgpu.ids | cut -c1-100; done
done; done
