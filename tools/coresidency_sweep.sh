set -e
cd $GRAFT_REPO_ROOT
out=gpurun_out/r3d_coresidency.jsonl
rm -f $out
for w in 0 5 4 3; do
  SMT_AGG_WAVES=$w python tools/coresidency_probe.py $out
  SMT_AGG_WAVES=$w SMT_HIP_LIB=$PWD/build/pf4/libsmt_hip.so python tools/coresidency_probe.py $out
done
SMT_AGG_WAVES=4 SMT_PIPE_SCHEDULE=2 SMT_HIP_LIB=$PWD/build/pf4/libsmt_hip.so python tools/coresidency_probe.py $out
SMT_AGG_WAVES=4 SMT_HIP_LIB=$PWD/build/pf6/libsmt_hip.so python tools/coresidency_probe.py $out
