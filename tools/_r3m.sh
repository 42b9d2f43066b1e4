cd $GRAFT_REPO_ROOT
python -m pytest tests/test_matchers_gpu.py tests/test_cpp_host_gpu.py -x -q -m gpu > gpurun_out/r3m_pytest.txt 2>&1; echo test_rc=$?
echo "impl 6 (slots)"; python tools/asw_run.py 2 6 | tail -1
echo "impl 3 (whole-image table)"; python tools/asw_run.py 2 3 | tail -1
python tools/asw_bisect.py gpurun_out/r3m_asw_bisect.json > gpurun_out/r3m_asw_bisect.log 2>&1; echo bisect_rc=$?
bash tools/prof_pmc_scanline.sh > gpurun_out/r3m_pmc.log 2>&1; echo pmc_rc=$?
