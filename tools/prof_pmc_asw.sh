# PMC passes over one ASW call (config-4 size), impl $1 (default 3): VALU / LDS / wait cycles of k_asw3.
set -x
IMPL=${1:-3}
O=gpurun_out/pmc_asw_$IMPL
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
A="python3 tools/asw_run.py 1 $IMPL"
rocprofv3 --pmc SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/a -o a -- $A > $O/a.out 2> $O/a.err
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS --kernel-trace --output-format csv -d $O/b -o b -- $A > $O/b.out 2> $O/b.err
rocprofv3 --pmc SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $O/c -o c -- $A > $O/c.out 2> $O/c.err
rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SMEM --kernel-trace --output-format csv -d $O/d -o d -- $A > $O/d.out 2> $O/d.err
python3 - <<PY
import csv, glob, collections
for f in sorted(glob.glob("$O/*/*counter_collection.csv")):
    acc = collections.defaultdict(float)
    for r in csv.DictReader(open(f)):
        if "k_asw3" in r["Kernel_Name"]:
            acc[r["Counter_Name"]] += float(r["Counter_Value"])
    n = sum(1 for r in csv.DictReader(open(f)) if "k_asw3" in r["Kernel_Name"])
    print(f.split("/")[-2], {k: v for k, v in acc.items()}, "rows", n)
PY
