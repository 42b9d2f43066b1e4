set -x
mkdir -p gpurun_out/pmc_agg
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
A="python3 tools/agg_ab.py --variants 6,7 --sws 8 --views L --reps 1"
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum --kernel-trace --output-format csv -d gpurun_out/pmc_agg/c -o c -- $A > gpurun_out/pmc_agg/a1.json 2> gpurun_out/pmc_agg/a1.err
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD --kernel-trace --output-format csv -d gpurun_out/pmc_agg/a -o a -- $A > gpurun_out/pmc_agg/a2.json 2> gpurun_out/pmc_agg/a2.err
rocprofv3 --pmc SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/pmc_agg/b -o b -- $A > gpurun_out/pmc_agg/a3.json 2> gpurun_out/pmc_agg/a3.err
rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM --kernel-trace --output-format csv -d gpurun_out/pmc_agg/d -o d -- $A > gpurun_out/pmc_agg/a4.json 2> gpurun_out/pmc_agg/a4.err
ls gpurun_out/pmc_agg/*/
