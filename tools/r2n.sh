set -x
mkdir -p gpurun_out/r2n
timeout -k 10 600 python -m pytest tests/test_matchers_gpu.py -x -q -m gpu -k asw > gpurun_out/r2n/pytest.txt 2>&1; tail -3 gpurun_out/r2n/pytest.txt
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r2n/p3 -o p -- python3 tools/asw_run.py 3 > gpurun_out/r2n/asw3.txt 2>&1
cat gpurun_out/r2n/asw3.txt | tail -2
timeout -k 10 100 python tools/asw_run.py 3
