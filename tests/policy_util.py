"""Helpers for the policy tests: real observations / masks from the CPU oracle, in the flat layouts of spec.py."""
import ctypes as C

import numpy as np
import torch


def oracle_batch_inputs(oracle, n=48, seed=3, steps=(0, 5, 40, 300, 900)):
    """-> dict of torch tensors (obs_f [B,1787], lists [B,5,25], lens [B,5], masks [B,325]) gathered at several game ages."""
    ob = oracle.OracleBatch(n, seed)
    F, Ls, Ln, M = [], [], [], []
    done = 0
    for s in steps:
        if s > done:
            ob.run_random(s - done, want_blobs=False); done = s
        f = np.zeros((n, 1787), dtype=np.float32); lists = np.zeros((n, 5, 25), dtype=np.int32)
        lens = np.zeros((n, 5), dtype=np.int32); pid = np.zeros((1,), dtype=np.int32)
        for i in range(n):
            ob.L.orc_obs(ob.env_ptr(i), f[i].ctypes.data_as(C.POINTER(C.c_float)), lists[i].ctypes.data_as(C.POINTER(C.c_int32)),
                         lens[i].ctypes.data_as(C.POINTER(C.c_int32)), pid.ctypes.data_as(C.POINTER(C.c_int32)))
        F.append(f); Ls.append(lists); Ln.append(lens); M.append(ob.masks())
    return dict(obs_f=torch.from_numpy(np.concatenate(F)), lists=torch.from_numpy(np.concatenate(Ls)),
                lens=torch.from_numpy(np.concatenate(Ln)), masks=torch.from_numpy(np.concatenate(M)))
