# FETCH_SIZE / WRITE_SIZE of the three scanline kernels and of the NCC kernels (SQ LDS counters), separate passes
# (TCC has 4 counter slots: FETCH_SIZE takes 3, WRITE_SIZE 2).  Output: gpurun_out/r3_pmc/*.json
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r3_pmc
mkdir -p $O
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/scan_fetch -o f -- python3 tools/scan_run.py 3 > $O/scan_f.out 2> $O/scan_f.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/scan_write -o w -- python3 tools/scan_run.py 3 > $O/scan_w.out 2> $O/scan_w.err
for D in 64 200; do
  rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/ncc_$D -o n -- python3 tools/ncc_run.py 1 2 $D > $O/ncc_$D.out 2> $O/ncc_$D.err
done
python3 tools/pmc_summary.py $O > $O/summary.json
cat $O/summary.json
