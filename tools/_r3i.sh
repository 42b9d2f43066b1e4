cd $GRAFT_REPO_ROOT
out=gpurun_out/r3i_coresidency_nt.jsonl
rm -f $out
export SMT_AGG_WAVES=0
python tools/coresidency_probe.py $out
for nt in 1 2 3; do SMT_HIP_LIB=$PWD/build/nt$nt/libsmt_hip.so python tools/coresidency_probe.py $out; done
