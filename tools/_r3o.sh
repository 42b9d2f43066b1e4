cd $GRAFT_REPO_ROOT
echo "impl 6"; python tools/asw_run.py 2 6 | tail -1
echo "impl 6 stagger 32"; SMT_HIP_LIB=$PWD/build/stag32/libsmt_hip.so python tools/asw_run.py 2 6 | tail -1
echo "impl 6 stagger 64"; SMT_HIP_LIB=$PWD/build/stag64/libsmt_hip.so python tools/asw_run.py 2 6 | tail -1
echo "impl 3"; python tools/asw_run.py 2 3 | tail -1
AGG_VARIANTS=12,7 SMT_AGG_WAVES=0 python tools/agg_time.py default12
