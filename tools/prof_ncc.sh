# kernel-trace stats (and PMC passes with "pmc" as $2) over one NCC call at 450x375, 21x21, D=$1
set -x
D=${1:-64}
O=gpurun_out/prof_ncc_$D
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
A="python3 tools/ncc_run.py 1 2 $D"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/s -o s -- $A > $O/s.out 2> $O/s.err
python3 - <<PY
import csv
for r in csv.DictReader(open("$O/s/s_kernel_stats.csv")):
    print(r["Name"][:70], r["Calls"], r["AverageNs"], r["MinNs"], r["MaxNs"])
PY
if [ "$2" = "pmc" ]; then
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/a -o a -- $A > $O/a.out 2> $O/a.err
rocprofv3 --pmc SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $O/b -o b -- $A > $O/b.out 2> $O/b.err
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d $O/c -o c -- $A > $O/c.out 2> $O/c.err
python3 - <<PY
import csv, glob, collections
for f in sorted(glob.glob("$O/[abc]/*counter_collection.csv")):
    acc = collections.defaultdict(float); n = collections.defaultdict(int)
    for r in csv.DictReader(open(f)):
        if "k_ncc2" in r["Kernel_Name"]:
            acc[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
    print({k: round(acc[k] / n[k]) for k in sorted(acc)})
PY
fi
