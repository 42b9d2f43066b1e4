set -x
mkdir -p gpurun_out/prof
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof/prof_stage -o stage -- python3 tools/stage_bench.py --size 1080p --reps 4 --matchers > gpurun_out/prof/stage_bench.json 2> gpurun_out/prof/stage.err
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof/prof_bench -o bench -- python3 bench.py --steps 200 --warmup 10 --cpu-rows 0 > gpurun_out/prof/bench_profiled.json 2> gpurun_out/prof/bench.err
find gpurun_out/prof -name "*kernel_stats*" | head; ls gpurun_out/prof/prof_stage | head
