# kernel trace of scanline beside aggregation, for SMT_AGG_WAVES = 0 (no limit) and 4
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for w in 0 4; do
  export SMT_AGG_WAVES=$w
  O=gpurun_out/r3e_trace_w$w
  mkdir -p $O
  rocprofv3 --kernel-trace --output-format csv -d $O -o t -- python3 tools/coresidency_trace.py 3 > $O/out.txt 2> $O/err.txt
  python3 - <<PY
import csv, glob
rows = []
for f in glob.glob("$O/**/*kernel_trace.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
rows = [r for r in rows if "k_scan" in r["Kernel_Name"] or "k_aggregate" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t0 = int(rows[0]["Start_Timestamp"])
print("SMT_AGG_WAVES=$w")
for r in rows:
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    print(f"{s/1e6:9.3f} -> {e/1e6:9.3f} ms  ({(e-s)/1e6:6.3f})  q{r.get('Queue_Id','?')}  {r['Kernel_Name'][:60]}  vgpr {r.get('VGPR_Count','?')} lds {r.get('LDS_Block_Size','?')}")
PY
done
./build/issue_probe > gpurun_out/r3e_issue_probe.txt 2>&1
