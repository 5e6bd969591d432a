export TMPDIR=/tmp
mkdir -p gpurun_out/r4
timeout 1500 python -m pytest tests/test_gpu_encode.py tests/test_golden.py tests/test_gpu_fused.py tests/test_gpu_fullsize.py tests/test_gpu_fuzz.py -x -q -m gpu 2>&1 | tail -3
for e in X=1 CLDN_HIP_SECTION_DV=0; do
env $e timeout 600 python bench.py --workload c3 --clouds 16 --steps 20 --warmup 5 --cpu-baseline-seconds 0 --e2e-seconds 0 --transcode-messages 0 --config-legs 0 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('$e c3', 'exact' if d.get('bit_exact') else 'WRONG', round(d['value']), round(d['repeats']['ms_per_step_median'],4), {k: round(v,4) for k,v in d['device_ms_per_step'].items()})"
env $e timeout 600 python tools/schemabench.py 2>&1 | grep -v amdgpu.ids | grep "ouster\|dds_sample\|step32"
done 2>&1 | tee gpurun_out/r4/t31.txt
