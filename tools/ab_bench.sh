#!/bin/bash
# A/B the cost kernel across alternative builds of libsmt_hip.so (build/variants/*.so); interleaved rounds.
for round in 1 2 3; do
  for v in default "$@"; do
    if [ $v = default ]; then unset SMT_HIP_LIB; else export SMT_HIP_LIB=$PWD/build/variants/libsmt_$v.so; fi
    python bench.py --cpu-rows 0 --steps 100 --warmup 10 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$v', 'round $round', 'pair_ms', d['ms_per_pair'], 'cost_ms', d['roofline']['kernel_ms'], 'tables_ms', d['roofline']['tables_ms'], 'frac', d['roofline']['frac'])"
  done
done
