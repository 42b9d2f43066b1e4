"""Deterministic synthetic stereo pairs: integer-only, platform independent.

LCG s <- s*1664525 + 1013904223 (mod 2^32), byte = s >> 24.  Vectorised with the
jump-ahead doubling identity so that full-HD pairs take milliseconds in numpy.
"""
import numpy as np

_A = 1664525
_C = 1013904223
_M = (1 << 32) - 1


def lcg_bytes(seed, n):
    """bytes of the n states FOLLOWING `seed`; returns (bytes uint8[n], last_state)."""
    out = np.empty(n, np.uint64)
    s = (int(seed) * _A + _C) & _M
    out[0] = s
    a, c, m = _A, _C, 1           # state[k+m] = a*state[k] + c  for the current block size m
    while m < n:
        k = min(m, n - m)
        out[m:m + k] = (out[:k] * np.uint64(a) + np.uint64(c)) & np.uint64(_M)
        c = (c * a + c) & _M      # compose the jump with itself: x -> a*(a*x+c)+c
        a = (a * a) & _M
        m *= 2
    return (out >> np.uint64(24)).astype(np.uint8), int(out[-1])


def _tri(x, p):
    m = x % (2 * p)
    return np.where(m < p, m, 2 * p - m) - p // 2


def synth_pair(H, W, D, seed, noise=False):
    """(L, R) uint8 [H][W]; same sequence as oracle/smt_oracle.c:orc_synth_pair."""
    b, st = lcg_bytes(seed, H * W)
    b = b.reshape(H, W).astype(np.int64)
    if noise:
        R = b
    else:
        jj = np.arange(W, dtype=np.int64)[None, :]
        ii = np.arange(H, dtype=np.int64)[:, None]
        # C integer division truncates toward zero; numerators may be negative
        t1 = _tri(jj, 203) * 70
        t2 = _tri(ii, 139) * 40
        q1 = np.sign(t1) * (np.abs(t1) // 101)
        q2 = np.sign(t2) * (np.abs(t2) // 69)
        R = 128 + q1 + q2 + 25 * (((jj // 40) + (ii // 30)) & 1) + (b % 6)
    R = np.clip(R, 0, 255).astype(np.uint8)
    b2, _ = lcg_bytes(st, H * W)
    b2 = b2.reshape(H, W)
    g = (D // 8 + ((np.arange(H) // 8) % 7) * (D // 16)).astype(np.int64)
    cols = np.arange(W, dtype=np.int64)[None, :] - g[:, None]
    Lsrc = np.take_along_axis(R, np.clip(cols, 0, W - 1), axis=1)
    L = np.where(cols >= 0, Lsrc, b2).astype(np.uint8)
    return L, R
