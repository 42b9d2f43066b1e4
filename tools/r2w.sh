mkdir -p gpurun_out/r2w
timeout -k 10 600 python -m pytest tests/test_matchers_gpu.py -x -q -m gpu -k asw 2>&1 | tail -3
timeout -k 10 100 python tools/asw_run.py 3
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r2w/p -o p -- python3 tools/asw_run.py 3 > /dev/null 2>&1
grep -h "k_asw" gpurun_out/r2w/p/p_kernel_stats.csv | awk -F'",' '{print substr($1,1,60), $2}' | cut -c1-140
