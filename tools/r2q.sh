set -x
mkdir -p gpurun_out/r2q
timeout -k 10 900 python -m pytest tests/test_config_hashes_gpu.py tests/test_adcensus_gpu.py tests/test_pipeline_gpu.py -x -q -m gpu > gpurun_out/r2q/pytest.txt 2>&1; tail -3 gpurun_out/r2q/pytest.txt
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --cpu-rows 0 > gpurun_out/r2q/bench.json 2> gpurun_out/r2q/bench.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r2q/bench.json").read().strip().splitlines()[-1])
r=d["roofline"]; print(d["value"], d["ms_per_pair"], r["kernel_ms"], r["frac"], r["store_ceiling_ms"], r["placement"]["candidate_pairs_tried"], r["store_mode"]["chosen"])
c=d["extra"]["configs"]["cfg3_pipeline_1080p_d192"]; print({k:v for k,v in c.items() if k!="stages"})
print(d["extra"].get("configs_error"))
PY
