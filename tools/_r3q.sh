cd $GRAFT_REPO_ROOT
python -m pytest tests -x -q -m gpu > gpurun_out/r3q_pytest.txt 2>&1; echo test_rc=$?
python bench.py --steps 20 --warmup 5 > gpurun_out/r3q_bench.json 2> gpurun_out/r3q_bench.err; echo bench_rc=$?
