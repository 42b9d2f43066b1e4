set -x
mkdir -p gpurun_out/r2o
timeout -k 10 600 python -m pytest tests/test_adcensus_gpu.py -x -q -m gpu > gpurun_out/r2o/pytest.txt 2>&1; tail -4 gpurun_out/r2o/pytest.txt
for m in auto nt plain; do
  if [ $m = auto ]; then unset SMT_STORE_MODE; else export SMT_STORE_MODE=$m; fi
  timeout -k 10 200 python bench.py --steps 50 --warmup 5 --cpu-rows 0 --no-extras 2>/dev/null | tail -1 > gpurun_out/r2o/bench_$m.json
  python - <<PY
import json
d=json.loads(open("gpurun_out/r2o/bench_$m.json").read()); r=d["roofline"]
print("$m", d["ms_per_pair"], r["kernel_ms"], r["frac"], r["store_ceiling_ms"], r["store_mode"], r["placement"]["candidate_pairs_tried"])
PY
done
