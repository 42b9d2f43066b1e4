set -x
mkdir -p gpurun_out/r2u
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/r2u/pytest.txt 2>&1; tail -3 gpurun_out/r2u/pytest.txt
timeout -k 10 300 python bench.py --steps 20 --warmup 5 > gpurun_out/r2u/bench.json 2> gpurun_out/r2u/bench.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r2u/bench.json").read().strip().splitlines()[-1])
r=d["roofline"]; print(d["value"], d["ms_per_pair"], r["kernel_ms"], r["frac"], r["store_ceiling_ms"], r["frac_of_store_ceiling"], r["sclk_mhz"], r["placement"]["candidate_pairs_tried"], r["store_mode"]["chosen"])
for k,v in d["extra"]["configs"].items():
    print(k, {a:b for a,b in v.items() if a!="stages"})
c=d["extra"]["configs"]["cfg3_pipeline_1080p_d192"]; print({k:v["ms"] for k,v in c["stages"].items()})
print(d.get("cpu_baseline"))
PY
