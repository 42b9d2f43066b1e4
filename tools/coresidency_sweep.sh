# Scanline beside aggregation, per aggregation occupancy (SMT_AGG_WAVES) and scanline prefetch depth.
# The prefetch-depth libraries are built in the build container first:
#   for pf in 4 6; do mkdir -p build/pf$pf; hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fvisibility=hidden \
#       -DSMT_SCAN_PF=$pf -c stereo_match_traditional_amd/csrc/scanline.hip -o build/pf$pf/scanline.o && \
#     hipcc --offload-arch=gfx950 -shared -fPIC $(ls stereo_match_traditional_amd/lib/obj/*.o | grep -v scanline) build/pf$pf/scanline.o \
#       -o build/pf$pf/libsmt_hip.so; done
set -e
cd $GRAFT_REPO_ROOT
out=gpurun_out/coresidency.jsonl
rm -f $out
for w in 0 5 4 3; do
  SMT_AGG_WAVES=$w python tools/coresidency_probe.py $out
  SMT_AGG_WAVES=$w SMT_HIP_LIB=$PWD/build/pf4/libsmt_hip.so python tools/coresidency_probe.py $out
done
SMT_AGG_WAVES=4 SMT_PIPE_SCHEDULE=2 SMT_HIP_LIB=$PWD/build/pf4/libsmt_hip.so python tools/coresidency_probe.py $out
SMT_AGG_WAVES=4 SMT_HIP_LIB=$PWD/build/pf6/libsmt_hip.so python tools/coresidency_probe.py $out
