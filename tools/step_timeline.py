"""Diagnostics (GPU box): the TIMELINE of one k_step launch inside the deferred loop - when every wave starts and ends, where it runs
and what type it is (catan_profile_enable(env, 3)).  Answers what the launch's duration consists of: the dispatch ramp (the spread of
start times), the waves themselves (by action type), waves that share a SIMD, the tail.  The per-type wave-time histogram VERDICT r4
asks for is the second table."""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from settlers_of_catan_rl_amd.env import VecCatanEnv
from settlers_of_catan_rl_amd import _lib

n = 65536
G = int(os.environ.get("CATAN_STEP_WAVE_GAMES", "32"))        # games per k_step wave (the library's default: 32)
FUSED = int(os.environ.get("FUSED", "1"))          # the fused-sampling loop (the default) / FUSED=0: sampler + k_step
env = VecCatanEnv(n, seed=0)
L = _lib.lib()
L.catan_set_deferred_fused(env.h, FUSED)
env.random_rollout_deferred(3000, 32)
L.catan_profile_enable(env.h, 3)
rows = max(n // 16 + 17, 7128)
BIN = ["settle", "road", "city", "buy_dev", "play_dev", "exchange", "propose", "respond", "robber", "roll", "end_turn", "steal", "discard",
       "play:1", "play:2", "play:3", "play:4", "no-op"]      # bins 0..12 = the action types in enum order (catan_state.h T_*, = the reference's ActionTypes), 13..16 = play_dev by card
spans, ramps, durs, by_bin, shared, tails, lates = [], [], [], {}, [], [], []
own_ramps, order_corr, own_spans = [], [], []
for rep in range(32):
    env.random_rollout_deferred(33 + rep, 32)
    out = np.zeros((rows, 8), dtype=np.uint32)
    L.catan_profile_read_waves(env.h, out.ctypes.data_as(C.c_void_p))
    a = out[:n // G + 17]
    rowidx = np.nonzero(a[:, 5] > 0)[0]
    a = a[a[:, 5] > 0]
    start = a[:, 2].astype(np.int64)
    # (rows of waves that this launch did not have keep an EARLIER launch's record: only the starts within 100 us of the latest belong)
    keep = start > start.max() - 10000
    a, start, rowidx = a[keep], start[keep], rowidx[keep]
    start = start - start.min()
    dur = a[:, [0, 1, 6, 7]].astype(np.int64).sum(1)
    end = start + dur
    spans.append((end.max() - start.min()) / 100.0)
    ramps.append(np.percentile(start, [50, 90, 99, 100]) / 100.0)
    durs.append(dur / 100.0)
    hw = a[:, 3]
    hw = hw.astype(np.int64)                        # HW_ID: wave slot in bits 0..3, then simd, pipe, cu, sh, se ...; XCC_ID in bits 28..31 here
    key = (hw >> 28) << 32 | ((hw >> 4) & 0xFFFFFF)
    uniq, cnt = np.unique(key, return_counts=True)
    shared.append((len(uniq), int((cnt > 1).sum()), int(cnt.max())))
    late = np.argsort(end)[-16:]
    tails.append([(BIN[int(a[i, 5]) - 1], start[i] / 100.0, dur[i] / 100.0) for i in late[::-1][:4]])
    lates.append(float((start > np.percentile(end, 5)).mean()))
    # the launch's OWN ramp: rows kept from an earlier launch (54 us before) shift the origin above, so measure the spread of the
    # starts around their median among the rows within 30 us of it, and the wave index (= dispatch order) against the start
    med = np.median(start)
    own = np.abs(start - med) < 3000
    so = start[own] - med
    own_ramps.append(np.percentile(so, [1, 10, 50, 90, 99]) / 100.0)
    idx = rowidx[own]
    order_corr.append(float(np.corrcoef(idx, so)[0, 1]))
    own_spans.append((end[own].max() - start[own].min()) / 100.0)
    for b in np.unique(a[:, 5]):
        sel = a[:, 5] == b
        by_bin.setdefault(int(b), []).append((a[sel][:, [0, 1, 6, 7]].astype(np.float64) / 100.0, start[sel] / 100.0))
L.catan_profile_enable(env.h, 0)
print(f"k_step timeline, 65 536 games, deferred W = 32, {'fused sampling' if FUSED else 'sampler + k_step'}; 32 launches; times in us (100 MHz wall clock)")
print(f"launch span (first wave start -> last wave end): mean {np.mean(spans):.2f}  min {np.min(spans):.2f}  max {np.max(spans):.2f}")
r = np.array(ramps)
print(f"wave START offsets: median {r[:, 0].mean():.2f}  p90 {r[:, 1].mean():.2f}  p99 {r[:, 2].mean():.2f}  last {r[:, 3].mean():.2f}")
d = np.concatenate(durs)
ro = np.array(own_ramps)
print(f"the launch's own waves (within 30 us of the median start): start - median at p1 {ro[:, 0].mean():.2f}  p10 {ro[:, 1].mean():.2f}  p90 {ro[:, 3].mean():.2f}  p99 {ro[:, 4].mean():.2f} us; "
      f"first start -> last end {np.mean(own_spans):.2f} us (min {np.min(own_spans):.2f}); correlation of wave index and start {np.mean(order_corr):.2f}")
print(f"wave durations: mean {d.mean():.2f}  p50 {np.percentile(d, 50):.2f}  p90 {np.percentile(d, 90):.2f}  p99 {np.percentile(d, 99):.2f}  max {d.max():.2f}")
print(f"waves that start after 5 % of the waves have already ended (a second round): {100 * np.mean(lates):.1f} %")
s = np.array(shared)
print(f"distinct (xcc, se, sh, cu, simd) used: {s[:, 0].mean():.0f}; SIMDs holding more than one wave: {s[:, 1].mean():.0f}; most waves on one SIMD: {s[:, 2].max()}")
print("the four waves that END last, per launch (type, start, duration): first 6 launches")
for t in tails[:6]:
    print("   " + "  ".join(f"{b}@{st:.1f}+{du:.1f}" for b, st, du in t))
print(f"{'type':10s} {'waves':>6s} {'stage-in':>9s} {'apply':>7s} {'masks':>7s} {'write':>7s} {'total':>7s} {'p99':>7s} {'start':>7s}")
for b in sorted(by_bin):
    ph = np.concatenate([x[0] for x in by_bin[b]]); st = np.concatenate([x[1] for x in by_bin[b]])
    tot = ph.sum(1)
    print(f"{BIN[b - 1]:10s} {len(tot) / 32:6.1f} {ph[:, 0].mean():9.2f} {ph[:, 1].mean():7.2f} {ph[:, 2].mean():7.2f} {ph[:, 3].mean():7.2f} {tot.mean():7.2f} {np.percentile(tot, 99):7.2f} {st.mean():7.2f}")
