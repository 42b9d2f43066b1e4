set -x
mkdir -p gpurun_out/r2f
python -m pytest tests -x -q -m gpu > gpurun_out/r2f/pytest.txt 2>&1; tail -8 gpurun_out/r2f/pytest.txt
python bench.py --steps 20 --warmup 5 > gpurun_out/r2f/bench_full.json 2> gpurun_out/r2f/bench_full.err
tail -c 600 gpurun_out/r2f/bench_full.err
