# the round's closing run on one GPU box: full -m gpu suite, the driver's bench command, the rocprofv3 kernel stats of
# bench.py and of the stage bench -> gpurun_out/final/
set -x
O=gpurun_out/final
mkdir -p $O
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
timeout -k 10 400 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_bench -o bench -- python3 bench.py --steps 20 --warmup 5 --cpu-rows 0 > $O/bench_profiled.json 2> $O/bench_profiled.err
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_stage -o stage -- python3 tools/stage_bench.py --size 1080p --reps 4 --matchers > $O/stage_bench.json 2> $O/stage.err
find $O -name "*kernel_stats*"
python - <<'PY'
import json
d=json.loads(open("gpurun_out/final/bench.json").read().strip().splitlines()[-1])
r=d["roofline"]; print(d["value"], d["ms_per_pair"], r["kernel_ms"], r["frac"], r["store_ceiling_ms"], r["frac_of_store_ceiling"], r["sclk_mhz"], r["placement"]["candidate_pairs_tried"], r["store_mode"]["chosen"])
print(d["extra"]["cfg5_kitti_256pairs_strong"])
for k,v in d["extra"]["configs"].items():
    print(k, {a:b for a,b in v.items() if a not in ("stages","note","bound","lds_source")})
c=d["extra"]["configs"]["cfg3_pipeline_1080p_d192"]; print({k:v["ms"] for k,v in c["stages"].items()})
print(d.get("cpu_baseline"))
PY
