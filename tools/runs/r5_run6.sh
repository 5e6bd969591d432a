export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_decode.py tests/test_gpu_fuzz.py tests/test_host_api.py tests/test_gpu_encode.py -x -q -m gpu 2>&1 | tail -4
export SCHEMABENCH_ONLY=xyz_rgb
for e in X=1 CLDN_HIP_FORM_KERNEL=1; do echo "== $e"; env $e bash tools/prof_any.sh form_$e python /root/repo/tools/schemabench.py 2>&1 | grep -v amdgpu | grep -i "k_dec\|k_mark\|decode\|encode" | cut -c1-200 | head -20; done
