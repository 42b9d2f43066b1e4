set -x
mkdir -p gpurun_out/r2e
timeout -k 10 300 python tools/agg_ab.py --variants 6,7 --sweeps 0 --sws 8,16,32 --reps 8 > gpurun_out/r2e/agg_ab4.json 2> gpurun_out/r2e/agg_ab4.err && cat gpurun_out/r2e/agg_ab4.json && timeout -k 10 400 python -m pytest tests/test_pipeline_gpu.py -x -q -m gpu -k "aggregation or cblsm or config3" 2>&1 | tail -3
