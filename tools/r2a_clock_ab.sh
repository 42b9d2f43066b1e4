set -x
mkdir -p gpurun_out/r2a
(rocm-smi --showclocks --showpower --showperflevel || true) > gpurun_out/r2a/smi_idle.txt 2>&1
cat /sys/class/drm/card*/device/pp_dpm_sclk > gpurun_out/r2a/sysfs_sclk.txt 2>&1 || true
cat /sys/class/drm/card*/device/pp_dpm_mclk > gpurun_out/r2a/sysfs_mclk.txt 2>&1 || true
for round in 1 2 3; do
  for cfg in "20 5" "200 10" "20 200" "2000 10"; do
    set -- $cfg
    python bench.py --cpu-rows 0 --steps $1 --warmup $2 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('steps $1 warmup $2 round $round', 'pair_ms', d['ms_per_pair'], 'cost_ms', d['roofline']['kernel_ms'], 'tables_ms', d['roofline']['tables_ms'], 'frac', d['roofline']['frac'])" | tee -a gpurun_out/r2a/ab.txt
  done
done
hipcc --offload-arch=gfx950 -O3 tools/store_ceiling.hip -o /tmp/sc && /tmp/sc > gpurun_out/r2a/store_ceiling.txt
(rocm-smi --showclocks --showpower || true) > gpurun_out/r2a/smi_after.txt 2>&1
