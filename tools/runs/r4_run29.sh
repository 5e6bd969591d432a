export TMPDIR=/tmp
timeout 600 python tools/schemabench.py 2>&1 | grep -v amdgpu.ids | grep -A1 "dds_sample_layout_step26"
CLDN_HIP_GORILLA_PREPASS=1 timeout 600 python tools/schemabench.py 2>&1 | grep -v amdgpu.ids | grep -A1 "dds_sample_layout_step26"
