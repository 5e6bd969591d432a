export TMPDIR=/tmp
mkdir -p gpurun_out/r4
timeout 1800 python -m pytest tests -x -q -m gpu 2>&1 | tail -8 > gpurun_out/r4/t13_tests.txt
cat gpurun_out/r4/t13_tests.txt
timeout 900 python bench.py > gpurun_out/r4/t13_bench.json 2> gpurun_out/r4/t13_bench.err
tail -c 3000 gpurun_out/r4/t13_bench.json
tail -5 gpurun_out/r4/t13_bench.err
