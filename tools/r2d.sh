set -x
mkdir -p gpurun_out/r2d
hipcc -w --offload-arch=gfx950 -O3 tools/alloc_probe.hip -o /tmp/ap && /tmp/ap > gpurun_out/r2d/alloc_probe.txt
for r in 1 2 3; do
  python bench.py --steps 20 --warmup 5 --cpu-rows 0 --no-extras 2>/dev/null | tail -1 >> gpurun_out/r2d/bench_place.jsonl
  SMT_PLACEMENT=0 python bench.py --steps 20 --warmup 5 --cpu-rows 0 --no-extras 2>/dev/null | tail -1 >> gpurun_out/r2d/bench_noplace.jsonl
done
python tools/agg_ab.py --variants 4 --sweeps 0,1 --sws 16,32,64,128 --reps 5 > gpurun_out/r2d/agg_ab.json 2> gpurun_out/r2d/agg_ab.err
python -m pytest tests/test_shard_gpu.py -x -q -m gpu > gpurun_out/r2d/pytest_shard.txt 2>&1
python bench.py --steps 20 --warmup 5 > gpurun_out/r2d/bench_full.json 2> gpurun_out/r2d/bench_full.err
tail -5 gpurun_out/r2d/pytest_shard.txt
