# A/B of two builds of the library on one box: tests, then bench lines alternating NEW (in-tree lib) / OLD ($OLD_LIB or tools/_old_libsmt_hip.so)
timeout -k 10 600 python -m pytest tests/test_adcensus_gpu.py tests/test_config_hashes_gpu.py -x -q -m gpu 2>&1 | tail -3
for i in 1 2; do
for w in adcensus_1080p_d192 adcensus_720p_d128 adcensus_kitti_d256; do
for which in NEW OLD; do
if [ $which = OLD ]; then export SMT_HIP_LIB=${OLD_LIB:-$PWD/tools/_old_libsmt_hip.so}; else unset SMT_HIP_LIB; fi
timeout -k 10 200 python bench.py --steps 40 --warmup 5 --no-extras --cpu-rows 0 --workload $w 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('$which', '$w', 'ms/pair', d['ms_per_pair'], 'kernel', r['kernel_ms'], 'ceiling', r.get('store_ceiling_ms'), 'sclk', r.get('sclk_mhz'), 'kcycles', round(r['kernel_ms']*(r.get('sclk_mhz') or 0)))"
done; done; done
