#!/usr/bin/env python3
"""Generates tests/golden/*.npz.  Run in the build container (needs /root/reference for the
CrossAggregator fixtures, whose expected outputs come from the reference's own
cross_aggregator.cpp compiled into oracle/_ref by oracle/Makefile).

The AD-Census LUT fixture records the two expf tables of this image's glibc so that a
different libm on a GPU node would be noticed.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import oracle as O  # noqa: E402


def main():
    O.build()
    assert O.have_ref(), "oracle/_ref missing: needs /root/reference"
    cases = [("a", 40, 56, 8, 7, False, (34, 17, 20, 6, 4)),
             ("b", 33, 47, 5, 11, True, (34, 17, 20, 6, 4)),
             ("c", 36, 90, 24, 3, False, (20, 9, 25, 8, 3))]
    for tag, H, W, D, seed, noise, prm in cases:
        L, _ = O.synth_pair(H, W, 16, seed, noise)
        bgr = O.synth_bgr(L, seed + 5)
        cost = (np.random.default_rng(seed).random((H, W, D), dtype=np.float32) * 2).astype(np.float32)
        arms, out = O.ref_crossagg(bgr, cost, *prm)
        np.savez_compressed(os.path.join(HERE, f"crossagg_{tag}.npz"), bgr=bgr, cost_init=cost, arms=arms,
                            cost_out=out, params=np.array(prm, np.int32))
    a, c = O.fuse_luts(10.0, 30.0)
    np.savez(os.path.join(HERE, "adcensus_luts_sc10_ss30.npz"), lutA=a, lutC=c)
    print("golden fixtures written")


if __name__ == "__main__":
    main()
