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


def get_obs_rows(vec, dtype=torch.float32, out=None, rows=None, t=None, sel=None, dense=True, games=None):
    """catan_obs_rows: the observations of all games in ONE pass, as the dense matrix for the policy (`out` = (f [n][1787] of
    `dtype` float32 / bfloat16, int32 lists, int32 lens) or None to allocate; dense=False: no dense output) and - for the games
    with sel[g] - appended to the rollout storage `rows` = (obs_f [steps][n][1787] of `dtype`, int8 lists [steps][n][5][25],
    int8 lens [steps][n][5]) at step t[g] (int64).  games (int32 [k], k <= n): catan_obs_rows_of - dense row j is game games[j]
    (only the first k rows of `out` are written) and only the listed games append.  -> (f, lists, lens) or None."""
    if dtype not in (torch.float32, torch.bfloat16):
        raise ValueError("observations are written as float32 or bfloat16")
    f = lists = lens = None
    if dense:
        if out is None:
            k = vec.n if games is None else games.numel()
            f = torch.empty((k, spec.OBS_FLOATS), dtype=dtype, device=vec.device)
            lists = torch.empty((k, 5, spec.OBS_LIST_PAD), dtype=torch.int32, device=vec.device)
            lens = torch.empty((k, 5), dtype=torch.int32, device=vec.device)
        else:
            f, lists, lens = out
            if f.dtype != dtype or lists.dtype != torch.int32 or lens.dtype != torch.int32 or not (f.is_contiguous() and lists.is_contiguous() and lens.is_contiguous()):
                raise ValueError("out = (f of the requested dtype, int32 lists, int32 lens), contiguous")
    rf = rl = rn = None
    if rows is not None:
        rf, rl, rn = rows
        if rf.dtype != dtype or rl.dtype != torch.int8 or rn.dtype != torch.int8 or rf.shape[1] != vec.n or not (rf.is_contiguous() and rl.is_contiguous() and rn.is_contiguous()):
            raise ValueError("rows = (obs_f [steps][n][1787] of the requested dtype, int8 lists, int8 lens), contiguous")
        t = t.to(torch.int64).contiguous()
        sel = (sel.view(torch.uint8) if sel.dtype == torch.bool else sel.to(torch.uint8)).contiguous()    # (bool is one byte, 0 / 1: no copy)
    p = lambda x: None if x is None else _ptr(x)
    if games is not None:
        if games.dtype != torch.int32 or not games.is_contiguous() or games.numel() > vec.n:
            raise ValueError("games = contiguous int32 game ids, at most n of them")
        _lib.check(vec.L.catan_obs_rows_of(vec.h, int(dtype == torch.bfloat16), p(f), p(lists), p(lens), p(rf), p(rl), p(rn), p(t) if rows is not None else None,
                                           p(sel) if rows is not None else None, p(games), games.numel(), _stream()))
        return (f, lists, lens) if dense else None
    _lib.check(vec.L.catan_obs_rows(vec.h, int(dtype == torch.bfloat16), p(f), p(lists), p(lens), p(rf), p(rl), p(rn), p(t) if rows is not None else None,
                                    p(sel) if rows is not None else None, _stream()))
    return (f, lists, lens) if dense else None


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
