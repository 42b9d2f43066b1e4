set -x
mkdir -p gpurun_out/r2e
python tools/agg_ab.py --variants 4,6 --sweeps 0 --sws 16 --reps 8 > gpurun_out/r2e/agg_ab.json 2> gpurun_out/r2e/agg_ab.err
python tools/agg_ab.py --variants 6,4 --sweeps 0 --sws 16 --reps 8 >> gpurun_out/r2e/agg_ab.json 2>> gpurun_out/r2e/agg_ab.err
cat gpurun_out/r2e/agg_ab.json
