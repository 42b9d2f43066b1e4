cd $GRAFT_REPO_ROOT
python -m pytest tests/test_pipeline_gpu.py -x -q -m gpu -k "test_aggregation and 10" > gpurun_out/r3j_pytest.txt 2>&1; echo test_rc=$?
AGG_VARIANTS=7,8,9,10,11 SMT_AGG_WAVES=0 python tools/agg_time.py mfma_forms > gpurun_out/r3j_agg.txt 2>&1
