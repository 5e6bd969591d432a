export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
export SCHEMABENCH_ONLY=xyz_rgb
for e in X=1 CLDN_HIP_STREAM_BITMAP=1; do echo "== $e"; env $e bash tools/prof_any.sh form_$e python /root/repo/tools/schemabench.py 2>&1 | grep -v amdgpu | grep -i "k_dec\|k_mark\|decode\|encode" | cut -c1-200 | head -20; done
