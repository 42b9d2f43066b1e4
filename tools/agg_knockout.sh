# Aggregation kernel: full build vs knock-out builds (-DSMT_AGG_KNOCKOUT=1: no flagged FMAs, =2: no tap loads).
# Build the two variant libraries first (in the build container; build/ travels to the GPU box):
#   for ko in 1 2; do mkdir -p build/ko$ko; hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fvisibility=hidden \
#       -DSMT_AGG_KNOCKOUT=$ko -c stereo_match_traditional_amd/csrc/crossarm.hip -o build/ko$ko/crossarm.o && \
#     hipcc --offload-arch=gfx950 -shared -fPIC $(ls stereo_match_traditional_amd/lib/obj/*.o | grep -v crossarm) build/ko$ko/crossarm.o \
#       -o build/ko$ko/libsmt_hip.so; done
# then on the GPU box:  bash tools/agg_knockout.sh
cd $GRAFT_REPO_ROOT
export AGG_VARIANTS=7,6 AGG_ROUNDS=1
for w in 0 4; do
  export SMT_AGG_WAVES=$w
  python tools/agg_time.py full
  SMT_HIP_LIB=$PWD/build/ko1/libsmt_hip.so python tools/agg_time.py no_fma
  SMT_HIP_LIB=$PWD/build/ko2/libsmt_hip.so python tools/agg_time.py no_loads
done
