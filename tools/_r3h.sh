cd $GRAFT_REPO_ROOT
python -m pytest tests/test_pipeline_gpu.py -x -q -m gpu -k "aggregation or cblsm_portrait or fuzz" > gpurun_out/r3h_pytest.txt 2>&1; echo test_rc=$?
AGG_VARIANTS=7,8,9 SMT_AGG_WAVES=0 python tools/agg_time.py mfma_vs_valu > gpurun_out/r3h_agg.txt 2>&1
AGG_VARIANTS=7,8,9 SMT_AGG_WAVES=4 python tools/agg_time.py mfma_vs_valu_w4 >> gpurun_out/r3h_agg.txt 2>&1
