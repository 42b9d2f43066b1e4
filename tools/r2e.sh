set -x
mkdir -p gpurun_out/r2e
python tools/agg_ab.py --variants 6,7 --sweeps 0 --sws 16,32 --reps 8 > gpurun_out/r2e/agg_ab2.json 2> gpurun_out/r2e/agg_ab2.err
cat gpurun_out/r2e/agg_ab2.json; tail -3 gpurun_out/r2e/agg_ab2.err
