set -x
mkdir -p gpurun_out/r2l
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/r2l/pytest.txt 2>&1; tail -4 gpurun_out/r2l/pytest.txt
timeout -k 10 300 python bench.py --steps 20 --warmup 5 > gpurun_out/r2l/bench.json 2> gpurun_out/r2l/bench.err
timeout -k 10 200 python tools/stage_bench.py --size 1080p --reps 4 > gpurun_out/r2l/stage.json 2> gpurun_out/r2l/stage.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r2l/bench.json").read().strip().splitlines()[-1])
r=d["roofline"]; print(d["value"], d["ms_per_pair"], r["kernel_ms"], r["frac"], r["store_ceiling_ms"], r["sclk_mhz"], r["placement"]["candidate_pairs_tried"])
c=d["extra"]["configs"]["cfg3_pipeline_1080p_d192"]; print(c["ms_per_pair"], c["frac_hbm_peak"], {k:v["ms"] for k,v in c["stages"].items()})
PY
