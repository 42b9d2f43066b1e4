# Counters of the default aggregation kernel (variant 13) beside variant 12, left view 1080p D=192, one pass per set.
# Output: gpurun_out/r3_pmc_agg/<set>/..., summary.json
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r3_pmc_agg
mkdir -p $O
rocprofv3 -L > $O/avail.txt 2>&1
A="python3 tools/agg_ab.py --variants 12,13 --sws 8 --views L --reps 1"
i=0
while read -r SET; do
  i=$((i+1))
  timeout -k 10 150 rocprofv3 --pmc $SET --kernel-trace --output-format csv -d $O/s$i -o p -- $A > $O/s$i.out 2> $O/s$i.err || echo "set $i failed: $SET"
done <<'SETS'
SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD
SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES GRBM_GUI_ACTIVE
SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM
SQ_INST_CYCLES_SALU SQ_INST_CYCLES_SMEM SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_ANY
SQ_WAVES SQ_LEVEL_WAVES SQ_INSTS_BRANCH SQ_ACTIVE_INST_MISC
SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQ_INSTS_SENDMSG
SQ_IFETCH SQ_IFETCH_LEVEL SQC_ICACHE_REQ SQC_ICACHE_MISSES
SQ_INST_LEVEL_SMEM SQ_INST_LEVEL_VMEM SQ_WAIT_INST_LDS SQ_INSTS_LDS
SETS
python3 tools/pmc_summary.py $O > $O/summary.json
cat $O/summary.json
