"""Decoding helpers for the fixtures written by tools/gen_golden.py."""
import os
import zlib

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


def crc(a):
    return zlib.crc32(np.ascontiguousarray(a).tobytes()) & 0xFFFFFFFF


def unpack_masks(packed):
    return np.unpackbits(packed, bitorder="little")[:325].astype(np.float32)


def decode_obs(traj):
    """-> list of float32[1787] observations at traj['sample_idx'] (stored as zero pattern + non-zero values)."""
    pat = np.unpackbits(traj["sample_obs"], axis=1)[:, :1787].astype(bool)
    vals = traj["sample_obs_nz"]
    out, k = [], 0
    for row in pat:
        o = np.zeros(1787, dtype=np.float32)
        n = int(row.sum())
        o[row] = vals[k:k + n]
        k += n
        out.append(o)
    assert k == len(vals)
    return out


# the last two were generated with EnvWrapper's NON-default keyword arguments (env/wrapper.py:12-13): dense rewards x
# env.reward_annealing_factor 0.37 with 1 proposed trade per turn; dense rewards with unlimited trades
# ; a finite max_actions_per_turn = 2 (wrapper.py:233-234: only EndTurn stays legal once actions_this_turn exceeds it)
TRAJS = ["traj_s3_e0.npz", "traj_s3_e1.npz", "traj_s17_e4.npz", "traj_dense037_t1_s5_e2.npz", "traj_dense_tnone_s5_e3.npz",
         "traj_maxact2_s7_e1.npz"]


def traj_kwargs(t):
    """-> (dense_reward, reward_annealing_factor, max_proposed_trades_per_turn, max_actions_per_turn) a trajectory was generated with"""
    ma = None if ("max_actions" not in t.files or int(t["max_actions"]) < 0) else int(t["max_actions"])
    if "dense" not in t.files:
        return False, 1.0, 4, ma
    tr = int(t["trades"])
    return bool(int(t["dense"])), float(t["anneal"]), (None if tr < 0 else tr), ma
