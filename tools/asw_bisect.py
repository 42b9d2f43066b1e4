#!/usr/bin/env python3
"""Bisects the round-2 wrong-data event (first smt_asw after an smt_ncc in a fresh process).
Every configuration below runs lib/matchers_main ONCE in its own fresh process (SAD -> NCC -> ASW left -> ASW right)
and compares the ASW maps with the oracle; SMT_ASW_VERIFY=1 additionally prints where the anchor tables differ
from their definition right after k_asw_anchor and again after k_asw3.  Writes one JSON record.
usage: python tools/asw_bisect.py out.json"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle as O   # noqa: E402  (diagnostic tool: the oracle is the checker)

H, W, D, seed = 40, 90, 32, 5
EXE = os.path.join(ROOT, "stereo_match_traditional_amd", "lib", "matchers_main")
CONFIGS = [
    ("default_pool_trimming__as_round2", {"SMT_SCRATCH_MODE": "default", "SMT_ASW_IMPL": "3"}),
    ("default_pool_trimming__verify", {"SMT_SCRATCH_MODE": "default", "SMT_ASW_IMPL": "3", "SMT_ASW_VERIFY": "1"}),
    ("default_pool_trimming__per_workgroup_slots", {"SMT_SCRATCH_MODE": "default", "SMT_ASW_IMPL": "6"}),
    ("default_pool_trimming__vector_loads", {"SMT_SCRATCH_MODE": "default", "SMT_ASW_IMPL": "5"}),
    ("default_pool_trimming__ncc_without_scratch", {"SMT_SCRATCH_MODE": "default", "SMT_ASW_IMPL": "3", "SMT_NCC_IMPL": "1"}),
    ("default_pool_trimming__asw_without_scratch", {"SMT_SCRATCH_MODE": "default", "SMT_ASW_IMPL": "1"}),
    ("plain_hipMalloc", {"SMT_SCRATCH_MODE": "malloc", "SMT_ASW_IMPL": "3"}),
    ("plain_hipMalloc__verify", {"SMT_SCRATCH_MODE": "malloc", "SMT_ASW_IMPL": "3", "SMT_ASW_VERIFY": "1"}),
    ("library_pool_never_trimming", {"SMT_SCRATCH_MODE": "pool", "SMT_ASW_IMPL": "3"}),
    ("library_pool_never_trimming__verify", {"SMT_SCRATCH_MODE": "pool", "SMT_ASW_IMPL": "3", "SMT_ASW_VERIFY": "1"}),
    ("arena_on_hipMalloc__shipped", {}),
    ("arena_on_hipMalloc__verify", {"SMT_ASW_VERIFY": "1"}),
]


def main():
    O.build()
    L, R = O.synth_pair(H, W, D, seed)
    Lp, Rp = O.pad_replicate(L, 4), O.pad_replicate(R, 4)
    sp, cm = O.asw_masks(3, 50.0, 30.0)
    exp = {"ncc": O.ncc(L, R, D, 3), "asw_left": O.asw(Lp, Rp, D, 3, sp, cm, 40, 0), "asw_right": O.asw(Lp, Rp, D, 3, sp, cm, 40, 1)}
    exp = {k: f"{O.fnv1a(v):016x}" for k, v in exp.items()}
    out = {"pair": [H, W, D, seed], "expected": exp, "runs": []}
    for name, env in CONFIGS:
        e = dict(os.environ)
        for k in ("SMT_SCRATCH_MODE", "SMT_ASW_VERIFY", "SMT_ASW_IMPL", "SMT_NCC_IMPL"):
            e.pop(k, None)
        e.update(env)
        r = subprocess.run([EXE, str(H), str(W), str(D), str(seed)], capture_output=True, text=True, timeout=300, env=e)
        got = dict(line.split() for line in r.stdout.strip().splitlines() if len(line.split()) == 2)
        rec = {"config": name, "env": env, "rc": r.returncode,
               "match": {k: got.get(k) == v for k, v in exp.items()},
               "verify": [l for l in r.stderr.splitlines() if l.startswith("SMT_ASW_VERIFY")]}
        out["runs"].append(rec)
        print(json.dumps(rec), flush=True)
    with open(sys.argv[1] if len(sys.argv) > 1 else "/dev/stdout", "w") as f:
        json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
