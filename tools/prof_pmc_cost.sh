# PMC passes over the headline cost kernel (k_cost_fast2<3,true,true>): instruction mix and busy fractions
set -x
O=gpurun_out/pmc_cost
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
A="python3 bench.py --steps 20 --warmup 5 --no-extras --cpu-rows 0"
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/a -o a -- $A > $O/a.out 2> $O/a.err
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $O/b -o b -- $A > $O/b.out 2> $O/b.err
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY --kernel-trace --output-format csv -d $O/c -o c -- $A > $O/c.out 2> $O/c.err
rocprofv3 --pmc SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d $O/d -o d -- $A > $O/d.out 2> $O/d.err
python3 - <<PY
import csv, glob, collections
for f in sorted(glob.glob("$O/[abcd]/*counter_collection.csv")):
    acc = collections.defaultdict(float); n = collections.defaultdict(int)
    for r in csv.DictReader(open(f)):
        if "k_cost_fast2<3, true, true>" in r["Kernel_Name"]:
            acc[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
    print({k: round(acc[k] / n[k]) for k in sorted(acc)}, {k: n[k] for k in n})
PY
tail -1 $O/a.out | cut -c1-300
