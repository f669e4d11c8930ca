#!/usr/bin/env python
"""CPU emulation of the front end's band-sum stage (mel.hip, round 5b): lane L owns the 8 consecutive bins [8L, 8L+8), bins with
the same triangle index j form a segment, the per-segment sums of the two slope contributions come out of a lane-local
recurrence + ONE wave-wide segmented scan of the lanes' open tails (DPP row_shr 1/2/4/8, row_bcast15, row_bcast31,
wave_shr 1) whose per-step predicates depend on the filterbank geometry only.  The emulation follows the kernel statement by
statement (same DPP lane semantics, same slot writes in the same order) and is compared with the definition

    band[b] = sum_{j_k == b} P_k u_k + sum_{j_k == b + 1} P_k (1 - u_k)

for random geometries (including ones whose segments span many lanes, empty triangles, everything out of range).
Used by tests/test_abi_cpu.py."""
import numpy as np

NC, W = 512, 64


def dpp(src, ctrl, row_mask=0xF):
    """v_mov_dpp with old = 0, bound_ctrl: lanes without a source or outside row_mask get 0"""
    out = np.zeros_like(src)
    for lane in range(W):
        row = lane >> 4
        if not (row_mask >> row) & 1:
            continue
        if 0x111 <= ctrl <= 0x11F:                   # row_shr:n
            n = ctrl - 0x110
            if (lane & 15) >= n:
                out[lane] = src[lane - n]
        elif ctrl == 0x138:                          # wave_shr:1
            if lane >= 1:
                out[lane] = src[lane - 1]
        elif ctrl == 0x142:                          # row_bcast15: lane 15 of each row to the next row
            if row >= 1:
                out[lane] = src[row * 16 - 1]
        elif ctrl == 0x143:                          # row_bcast31: lane 31 to rows 2 and 3
            if row >= 2:
                out[lane] = src[31]
        else:
            raise ValueError(hex(ctrl))
    return out


STEPS = [(0x111, 0xF), (0x112, 0xF), (0x114, 0xF), (0x118, 0xF), (0x142, 0xA), (0x143, 0xC)]


def lane_constants(J, n_mels):
    """what the kernel's prologue derives from sJ: per lane and bin keep / slot address, per lane and scan step the predicate"""
    DUMMY0 = n_mels + 2                              # one private dummy slot per lane behind the n_mels + 1 real ones (+1 pad)
    keep = np.ones((W, 8), bool)
    addr = np.zeros((W, 9), int)
    for lane in range(W):
        for i in range(8):
            k = 8 * lane + i
            boundary = k > 0 and J[k] != J[k - 1]
            keep[lane, i] = not boundary
            jp = J[k - 1] if k > 0 else -1
            addr[lane, i] = jp if (boundary and 0 <= jp <= n_mels) else DUMMY0 + lane
        jl = J[NC - 1]
        addr[lane, 8] = jl if (lane == W - 1 and 0 <= jl <= n_mels) else DUMMY0 + lane
    f = (~keep).any(axis=1).astype(np.int64)         # the lane's tail starts a new segment
    wmask = np.zeros((len(STEPS), W), bool)
    for s, (ctrl, rm) in enumerate(STEPS):
        wmask[s] = f == 0
        f = f | dpp(f, ctrl, rm)
    return keep, addr, wmask, DUMMY0 + W


def band_sums(P, U, J, n_mels):
    """P[512] power, U[512] up-slope weights, J[512] triangle indices (non-decreasing) -> band[n_mels]"""
    keep, addr, wmask, nslots = lane_constants(J, n_mels)
    Pl = P.reshape(W, 8).astype(np.float32)
    Ul = U.reshape(W, 8).astype(np.float32)
    up = Pl * Ul
    dn = Pl - up
    val = np.stack([up, dn], axis=-1)                # [lane][i][2]
    slots = np.full((nslots, 2), np.nan, np.float32)
    slots[:n_mels + 2] = 0                           # the wave zeroes the real slots
    # pass A: tail of every lane
    acc = np.zeros((W, 2), np.float32)
    for i in range(8):
        acc = np.where(keep[:, i, None], acc, 0) + val[:, i]
    # segmented inclusive scan of the tails
    t = acc.copy()
    for s, (ctrl, rm) in enumerate(STEPS):
        sh = np.stack([dpp(t[:, 0], ctrl, rm), dpp(t[:, 1], ctrl, rm)], axis=-1)
        t = t + np.where(wmask[s][:, None], sh, 0)
    carry = np.stack([dpp(t[:, 0], 0x138), dpp(t[:, 1], 0x138)], axis=-1)
    # pass B: flush every finished segment
    acc = carry
    for i in range(8):
        slots[addr[:, i]] = acc                      # all lanes write (most into their dummy)
        acc = np.where(keep[:, i, None], acc, 0) + val[:, i]
    slots[addr[:, 8]] = acc
    b = np.arange(n_mels)
    return slots[b, 0] + slots[b + 1, 1]


def band_sums_definition(P, U, J, n_mels):
    out = np.zeros(n_mels, np.float64)
    for k in range(NC):
        j = J[k]
        if 0 <= j < n_mels:
            out[j] += float(P[k]) * float(U[k])
        if 1 <= j <= n_mels:
            out[j - 1] += float(P[k]) * (1.0 - float(U[k]))
    return out


def random_geometry(rng, n_mels, kind):
    if kind == "kaldi":                              # the shape of the real thing: mel-spaced triangles over linear bins
        fmin, fmax = rng.uniform(0, 10), rng.uniform(14000, 16000)
        mel = lambda f: 1127.0 * np.log1p(f / 700.0)
        t = (mel(np.arange(NC) * 31.25) - mel(fmin)) / ((mel(fmax) - mel(fmin)) / (n_mels + 1))
    elif kind == "wide":                             # few triangles: segments span many lanes
        t = np.sort(rng.uniform(-2, n_mels + 3, NC))
    elif kind == "out":                              # everything below / above the bank
        t = np.full(NC, -5.0) if rng.random() < 0.5 else np.full(NC, n_mels + 7.0)
    else:                                            # steps at random places, with gaps (empty triangles)
        t = np.cumsum(rng.choice([0, 0, 0, 0.3, 1.0, 2.5], NC)) - 1.5
    fl = np.floor(t)
    J = np.clip(fl, -1, 100000).astype(np.int64)
    return J, (t - fl).astype(np.float32)


def self_check(trials=60, seed=0):
    rng = np.random.default_rng(seed)
    worst = 0.0
    for it in range(trials):
        n_mels = int(rng.choice([128, 128, 64, 40, 4, 5]))
        J, U = random_geometry(rng, n_mels, ["kaldi", "wide", "out", "steps"][it % 4])
        P = (rng.random(NC).astype(np.float32) ** 4) * 10
        got = band_sums(P, U, J, n_mels)
        ref = band_sums_definition(P, U, J, n_mels)
        assert np.isfinite(got).all(), (it, "a band read a slot nobody wrote")
        err = np.abs(got - ref).max() / max(np.abs(ref).max(), 1e-30)
        worst = max(worst, err)
        assert err < 5e-6, (it, n_mels, err)
    return worst


if __name__ == "__main__":
    print("worst relative error over the trials:", self_check())
