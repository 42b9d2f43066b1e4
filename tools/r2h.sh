set -x
mkdir -p gpurun_out/r2h
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d gpurun_out/r2h/prof_stage -o stage -- python3 tools/stage_bench.py --size 1080p --reps 4 --matchers > gpurun_out/r2h/stage_bench.json 2> gpurun_out/r2h/stage.err
rocprofv3 --kernel-trace --stats -d gpurun_out/r2h/prof_bench -o bench -- python3 bench.py --steps 200 --warmup 10 --cpu-rows 0 > gpurun_out/r2h/bench_profiled.json 2> gpurun_out/r2h/bench.err
find gpurun_out/r2h -name "*kernel_stats*" | head; ls gpurun_out/r2h/prof_stage | head
