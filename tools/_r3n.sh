cd $GRAFT_REPO_ROOT
python -m pytest tests/test_matchers_gpu.py -x -q -m gpu -k "asw or scratch" > gpurun_out/r3n_pytest.txt 2>&1; echo test_rc=$?
echo "impl 6 (slots)"; python tools/asw_run.py 2 6 | tail -1
echo "impl 3 (whole-image table)"; python tools/asw_run.py 2 3 | tail -1
echo "impl 6 (slots)"; python tools/asw_run.py 2 6 | tail -1
