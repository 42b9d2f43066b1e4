set -x
mkdir -p gpurun_out/r2g
python -m pytest tests -x -q -m gpu > gpurun_out/r2g/pytest.txt 2>&1; tail -8 gpurun_out/r2g/pytest.txt
hipcc -w --offload-arch=gfx950 -O3 tools/store_ceiling.hip -o /tmp/sc && /tmp/sc > gpurun_out/r2g/store_ceiling.txt
hipcc -w --offload-arch=gfx950 -O3 tools/alloc_probe.hip -o /tmp/ap && /tmp/ap > gpurun_out/r2g/alloc_probe.txt
python bench.py --steps 20 --warmup 5 --cpu-rows 0 > gpurun_out/r2g/bench.json 2> gpurun_out/r2g/bench.err
head -8 gpurun_out/r2g/store_ceiling.txt; head -14 gpurun_out/r2g/alloc_probe.txt | cut -c1-60,100-200
