set -x
mkdir -p gpurun_out/r2m
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
A="python3 tools/asw_run.py 1"
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD --kernel-trace --output-format csv -d gpurun_out/r2m/a -o a -- $A > gpurun_out/r2m/a1.txt 2> gpurun_out/r2m/a1.err
rocprofv3 --pmc SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/r2m/b -o b -- $A > gpurun_out/r2m/a2.txt 2> gpurun_out/r2m/a2.err
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAVE_CYCLES --kernel-trace --output-format csv -d gpurun_out/r2m/c -o c -- $A > gpurun_out/r2m/a3.txt 2> gpurun_out/r2m/a3.err
cat gpurun_out/r2m/a1.txt
