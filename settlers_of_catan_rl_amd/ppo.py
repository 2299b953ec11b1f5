"""GAE and clipped-PPO loss on the HIP kernels (csrc/catan_ppo.hip), mirroring the reference's
BatchProcessor.compute_advantages_alt (RL/ppo/process_batch.py:134-142) and PPO.update loss (RL/ppo/ppo.py:46-66).
torch is plumbing: device buffers, streams, autograd hookup and (for N>1 ranks) the 3-scalar all-reduce."""
import ctypes as C

import torch

from . import _lib


def _ptr(t):
    return C.c_void_p(t.data_ptr())


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _hip_gae_raw(r, v, m, gamma, gae_lambda):
    """k_gae + k_adv_stats: -> (returns, adv_raw, stats3 = (sum, sum of squares, count) of adv_raw as float64[3])"""
    L = _lib.lib()
    T, N = r.shape
    returns = torch.empty_like(r)
    adv = torch.empty_like(r)
    ws = torch.empty((L.catan_gae_workspace_doubles(N),), dtype=torch.float64, device=r.device)
    stats = torch.empty((3,), dtype=torch.float64, device=r.device)
    _lib.check(L.catan_gae(_ptr(r), _ptr(v), _ptr(m), T, N, float(gamma), float(gae_lambda), _ptr(returns), _ptr(adv),
                           _ptr(ws), _ptr(stats), _stream()))
    return returns, adv, stats


def _hip_adv_normalise(adv, stats):
    """k_adv_normalise, in place"""
    _lib.check(_lib.lib().catan_adv_normalise(_ptr(adv), adv.numel(), _ptr(stats), _stream()))
    return adv


# the two device back-ends of compute_gae; the world_size-2 gloo test swaps in torch stand-ins so that the function's own
# distributed logic (what is reduced, over which group, in which order) runs on CPU
_gae_raw, _adv_normalise = _hip_gae_raw, _hip_adv_normalise


def compute_gae(rewards, values, masks, gamma=0.999, gae_lambda=0.95, process_group=None, normalise=True):
    """rewards [T,N], values [T+1,N] (denormalised), masks [T+1,N] float32 CUDA -> (returns [T,N], advantages [T,N]).
    When torch.distributed is initialised the advantage mean/std are global over all ranks (three doubles
    all-reduced over `process_group`, default group if None); pass process_group=False to keep them rank-local."""
    T, N = rewards.shape
    assert values.shape == (T + 1, N) and masks.shape == (T + 1, N)
    r, v, m = (x.contiguous().float() for x in (rewards, values, masks))
    returns, adv, stats = _gae_raw(r, v, m, gamma, gae_lambda)
    if normalise:
        dist = torch.distributed
        if process_group is not False and dist.is_available() and dist.is_initialized() and dist.get_world_size(process_group) > 1:
            dist.all_reduce(stats, group=process_group)     # (sum, sumsq, count) over all ranks: global mean / std
        adv = _adv_normalise(adv, stats)
    return returns, adv


_LOSS_WS = {}


def _loss_workspace(device):
    """zeroed once; every k_ppo_loss call leaves it zero (include/catan_hip.h).  One per (device, STREAM): launches in flight on
    two streams (the trainer beside the reference_api.PPO adapter, a side-stream caller) must not share partial sums and the
    arrival counter."""
    key = (device.type, device.index, torch.cuda.current_stream(device).cuda_stream if device.type == "cuda" else 0)
    if key not in _LOSS_WS:
        _LOSS_WS[key] = torch.zeros((_lib.lib().catan_ppo_loss_workspace_doubles(),), dtype=torch.float64, device=device)
    return _LOSS_WS[key]


class _PpoLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logp, values, old_logp, adv, old_values, returns, clip, value_coef, norm):
        L = _lib.lib()
        B = logp.numel()
        args = [x.contiguous().float().view(-1) for x in (logp, old_logp, adv, values, old_values, returns)]
        losses = torch.empty((2,), dtype=torch.float32, device=logp.device)
        d_logp = torch.empty((B,), dtype=torch.float32, device=logp.device)
        d_values = torch.empty((B,), dtype=torch.float32, device=logp.device)
        use_norm, mean, std = (0, 0.0, 1.0) if norm is None else (1, float(norm[0]), float(norm[1]))
        _lib.check(L.catan_ppo_loss(*[_ptr(x) for x in args], B, float(clip), float(value_coef), use_norm, mean, std,
                                    _ptr(losses), _ptr(d_logp), _ptr(d_values), _ptr(_loss_workspace(logp.device)), _stream()))
        ctx.save_for_backward(d_logp, d_values)
        ctx.shapes = (logp.shape, values.shape)
        total = losses[1] * value_coef + losses[0]
        ctx.mark_non_differentiable(losses)
        return total, losses

    @staticmethod
    def backward(ctx, g_total, _g_losses):
        d_logp, d_values = ctx.saved_tensors
        return (g_total * d_logp).view(ctx.shapes[0]), (g_total * d_values).view(ctx.shapes[1]), None, None, None, None, None, None, None


def _hip_ppo_loss(action_log_probs, values, old_action_log_probs, adv_targets, value_preds, returns, clip_param, value_loss_coef,
                  value_normaliser):
    return _PpoLoss.apply(action_log_probs, values, old_action_log_probs, adv_targets, value_preds, returns,
                          clip_param, value_loss_coef, value_normaliser)


_loss_backend = _hip_ppo_loss        # (replaceable like the GAE back-ends above)


def ppo_loss(action_log_probs, values, old_action_log_probs, adv_targets, value_preds, returns, clip_param=0.2,
             value_loss_coef=1.0, value_normaliser=None):
    """-> (value_loss_coef * value_loss + action_loss, (action_loss, value_loss)).  `value_normaliser` = (mean, std)
    applies RL/models/utils.py:17-18 to value_preds and returns first, as RL/ppo/ppo.py:46-48 does."""
    return _loss_backend(action_log_probs, values, old_action_log_probs, adv_targets, value_preds, returns,
                         clip_param, value_loss_coef, value_normaliser)
