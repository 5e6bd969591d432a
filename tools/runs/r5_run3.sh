export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_fuzz.py -x -q -m gpu -k "very_wide or wide_schema" 2>&1 | tail -25
