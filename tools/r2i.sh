set -x
mkdir -p gpurun_out/r2i
python -m pytest tests/test_pipeline_gpu.py tests/test_matchers_gpu.py tests/test_config_hashes_gpu.py -x -q -m gpu -k "arm or option or config3" > gpurun_out/r2i/pytest.txt 2>&1; tail -5 gpurun_out/r2i/pytest.txt
python - <<'PY' > gpurun_out/r2i/arms.txt 2>&1
import sys, numpy as np, torch
sys.path.insert(0, ".")
import stereo_match_traditional_amd as smt
from stereo_match_traditional_amd import synth
DEV = torch.device("cuda:0")
H, W, D = 1080, 1920, 192
L, R = synth.synth_pair(H, W, D, 3)
Lu = torch.from_numpy(L).to(DEV)
ca = smt.CrossArmAggregation().Initialize(H, W, 30, D, DEV)
def timed(fn, reps=20):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps
for walk in (True, False, True, False):
    ca.set_arm_walk(walk)
    print("walk" if walk else "masks", round(timed(lambda: ca.ComputeArmLengths(Lu)), 4), "ms per image")
PY
cat gpurun_out/r2i/arms.txt
