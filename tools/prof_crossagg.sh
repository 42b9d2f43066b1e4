# kernel-trace stats (+ optional PMC passes with "pmc" as $1) over CrossAggregator at 720p D=128
set -x
O=gpurun_out/prof_crossagg
mkdir -p $O
timeout -k 10 100 python tools/crossagg_run.py 2
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
A="python3 tools/crossagg_run.py 1"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/s -o s -- $A > $O/s.out 2> $O/s.err
python3 - <<PY
import csv
for r in csv.DictReader(open("$O/s/s_kernel_stats.csv")):
    print(r["Name"][28:75], r["Calls"], r["AverageNs"], r["MinNs"], r["MaxNs"])
PY
if [ "$1" = "pmc" ]; then
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum --kernel-trace --output-format csv -d $O/a -o a -- $A > $O/a.out 2> $O/a.err
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/b -o b -- $A > $O/b.out 2> $O/b.err
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d $O/c -o c -- $A > $O/c.out 2> $O/c.err
python3 - <<PY
import csv, glob, collections, re
for f in sorted(glob.glob("$O/[abc]/*counter_collection.csv")):
    acc = collections.defaultdict(float); n = collections.defaultdict(int)
    for r in csv.DictReader(open(f)):
        m = re.search(r"k_ca_pass2<([^>]*)>", r["Kernel_Name"])
        if m:
            key = (m.group(1), r["Counter_Name"])
            acc[key] += float(r["Counter_Value"]); n[key] += 1
    for k in sorted(acc): print(k, round(acc[k] / n[k]), n[k])
PY
fi
