set -x
mkdir -p gpurun_out/pmc
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
B="python3 bench.py --steps 10 --warmup 2 --cpu-rows 0 --no-extras"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc/pmc_fetch -o f -- $B > gpurun_out/pmc/b1.json 2> gpurun_out/pmc/b1.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc/pmc_write -o w -- $B > gpurun_out/pmc/b2.json 2> gpurun_out/pmc/b2.err
A="python3 tools/agg_ab.py --variants 6 --sws 16 --views L,R --reps 1"
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum --kernel-trace --output-format csv -d gpurun_out/pmc/agg_c -o c -- $A > gpurun_out/pmc/a1.json 2> gpurun_out/pmc/a1.err
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD --kernel-trace --output-format csv -d gpurun_out/pmc/agg_a -o a -- $A > gpurun_out/pmc/a2.json 2> gpurun_out/pmc/a2.err
rocprofv3 --pmc SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/pmc/agg_b -o b -- $A > gpurun_out/pmc/a3.json 2> gpurun_out/pmc/a3.err
find gpurun_out/pmc -name "*.csv" | head -20
