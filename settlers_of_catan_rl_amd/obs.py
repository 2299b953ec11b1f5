"""Observation encoder binding (csrc/catan_obs.hip): batched EnvWrapper._get_obs (reference env/wrapper.py:52-83)."""
import ctypes as C

import numpy as np
import torch

from . import _lib, spec


def _ptr(t):
    return C.c_void_p(t.data_ptr())


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def get_obs(vec, out=None):
    """-> (float32 [n][1787], int32 [n][5][25] card-id lists, int32 [n][5] lengths).  Pass `out` (same tuple) to reuse
    buffers, e.g. a [T+1][n] slice of the rollout storage."""
    if out is None:
        f = torch.empty((vec.n, spec.OBS_FLOATS), dtype=torch.float32, device=vec.device)
        lists = torch.empty((vec.n, 5, spec.OBS_LIST_PAD), dtype=torch.int32, device=vec.device)
        lens = torch.empty((vec.n, 5), dtype=torch.int32, device=vec.device)
    else:
        f, lists, lens = out
    _lib.check(vec.L.catan_obs(vec.h, _ptr(f), _ptr(lists), _ptr(lens), _stream()))
    return f, lists, lens


def obs_dict(vec, f, lists, lens):
    """Split the flat encoder output into the reference's observation keys (RL/ppo/process_batch.py:10-13), batched:
    'normal' keys -> float32 [n, ...], 'list' keys -> int64 [n, 25] zero padded (the net masks by length)."""
    out = {}
    for k, shp in spec.OBS_FLOAT_KEYS.items():
        o = spec.OBS_FLOAT_OFFSETS[k]
        n = int(np.prod(shp))
        out[k] = f[:, o:o + n].reshape((f.shape[0],) + shp)
    for i, k in enumerate(spec.OBS_LIST_KEYS):
        out[k] = lists[:, i].to(torch.int64)
        out[k + "_len"] = lens[:, i]
    return out


def single_env_obs(vec):
    """The reference's per-env observation dict (numpy), for the EnvWrapper shim."""
    f, lists, lens = get_obs(vec)
    pid = vec.deciding_player()
    f = f[0].cpu().numpy(); lists = lists[0].cpu().numpy(); lens = lens[0].cpu().numpy()
    obs = {"player_id": int(pid[0].item())}
    for k, shp in spec.OBS_FLOAT_KEYS.items():
        o = spec.OBS_FLOAT_OFFSETS[k]
        n = int(np.prod(shp))
        v = f[o:o + n].reshape(shp).astype(np.float64)
        obs[k] = [row for row in v] if k == "tile_representations" else v
    for i, k in enumerate(spec.OBS_LIST_KEYS):
        obs[k] = lists[i, :lens[i]].astype(np.int64)
    return obs
