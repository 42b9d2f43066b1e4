# aggregation kernel: full build vs knock-out builds (1: no flagged FMAs, 2: no tap loads), waves per SIMD unlimited and 4
cd $GRAFT_REPO_ROOT
export AGG_VARIANTS=7,6
for w in 0 4; do
  export SMT_AGG_WAVES=$w
  python tools/agg_time.py full
  SMT_HIP_LIB=$PWD/build/ko1/libsmt_hip.so python tools/agg_time.py no_fma
  SMT_HIP_LIB=$PWD/build/ko2/libsmt_hip.so python tools/agg_time.py no_loads
done
