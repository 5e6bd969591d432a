export TMPDIR=/tmp
for n in 1 2 4 8; do echo -n "clouds=$n "; FINBENCH_CLOUDS=$n timeout 300 python tools/finbench.py 2>&1 | grep -v amdgpu.ids; done
for s in 1 2 4 8; do echo -n "clouds=1 splits=$s "; CLDN_HIP_FINISH_SPLITS=$s FINBENCH_CLOUDS=1 timeout 300 python tools/finbench.py 2>&1 | grep -v amdgpu.ids; done
echo -n "clouds=1 T512 "; CLDN_HIP_FINISH_1024_BELOW=0 FINBENCH_CLOUDS=1 timeout 300 python tools/finbench.py 2>&1 | grep -v amdgpu.ids
bash tools/prof_any.sh r4_one env FINBENCH_CLOUDS=1 python /root/repo/tools/finbench.py 2>&1 | cut -c1-150 | tail -12
