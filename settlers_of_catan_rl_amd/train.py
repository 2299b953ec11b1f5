"""PPO update on device: the loop of `PPO.update` (reference RL/ppo/ppo.py:25-79) over the storage written by
`rollout.RolloutCollector`, with the reference's defaults (RL/ppo/arguments.py:4-116).

Per epoch (ppo.py:30-32, recompute_returns): values of all (T+1)*N stored observations are re-evaluated in chunks
(process_batch.py:108-132), GAE + advantage normalisation run on the HIP kernels (ppo.compute_gae; global statistics
over ranks), then `num_mini_batch` random minibatches of T*N // num_mini_batch rows (process_batch.py:169-200):
evaluate_actions -> fused PPO loss kernel (fwd+bwd) -> gradient all-reduce (one flat bucket over RCCL when N>1 ranks)
-> clip_grad_norm_(0.5) -> Adam.  The network runs under bf16 autocast on the GPU.

With an LSTM policy (`include_lstm`, off in the reference's defaults) the minibatches are the truncated-BPTT sequences of
`generator_lstm` (process_batch.py:203-293): every game's T stored decisions are cut into T / truncated_seq_len
consecutive pieces, a minibatch is a random set of pieces laid out time-major, entered with the LSTM state stored at the
piece's first decision and carrying the terminal masks inside; the value re-evaluation is one LSTM step per stored
decision from its stored state (process_batch.py:117-121, the `x.size(0) == hxs.size(0)` branch of policy.py:117-123).
"""
import sys
import time

import torch

from . import dist as cdist
from . import ppo as ppo_kernels
from . import nn_kernels


class PPOConfig(object):
    # RL/ppo/arguments.py defaults
    lr = 3e-4
    eps = 1e-5
    gamma = 0.999
    gae_lambda = 0.95
    clip_param = 0.2
    ppo_epoch = 10
    num_mini_batch = 64
    value_loss_coef = 1.0
    entropy_coef = 0.04
    max_grad_norm = 0.5
    truncated_seq_len = 10            # arguments.py:57-59 (LSTM policies only)
    value_chunk = 1048576      # rows per forward of the value pass (swept on MI355X at 13.2 M rows: 131 072: 184 ms, 262 144: 177, 524 288: 171, 1 048 576: 164; ~4 GB of activations)

    def __init__(self, **kw):
        for k, v in kw.items():
            if not hasattr(self, k):
                raise AttributeError(k)
            setattr(self, k, v)


def lstm_minibatches(T, N, seq_len, num_mini_batch, perm):
    """The index arithmetic of `generator_lstm` (process_batch.py:203-241).  Pieces are numbered game-major
    (piece k = game k // (T/seq_len), first decision (k % (T/seq_len)) * seq_len, :211-215); `perm` is the random
    permutation of the pieces (:216).  Yields per minibatch (t [seq_len, n], game [n]): decision (t[l, j], game[j]) is row
    l*n + j of the flattened batch (`_flatten_helper`, :274-291); the LSTM state going in is the one stored at (t[0, j], game[j])."""
    if T % seq_len != 0:
        raise ValueError("num_steps must be a multiple of truncated_seq_len (process_batch.py:206)")
    pieces = T // seq_len
    n = (T * N) // num_mini_batch // seq_len                                  # num_sequences_per_minibatch (:208)
    if n < 1 or (N * pieces) % n != 0:
        raise ValueError("the sequences do not divide into equal minibatches (the reference's .view(N, -1) at :257 needs that)")
    steps = torch.arange(seq_len, device=perm.device)
    for s in range(0, N * pieces, n):
        k = perm[s:s + n]
        game, t0 = k // pieces, (k % pieces) * seq_len
        yield t0[None, :] + steps[:, None], game


class PPOTrainer(object):
    def __init__(self, policy, cfg=None, autocast_dtype=torch.bfloat16, seed=0):
        self.policy, self.cfg = policy, cfg or PPOConfig()
        self.autocast_dtype = autocast_dtype
        if next(policy.parameters()).is_cuda:
            from . import nn_kernels
            nn_kernels.use_tuned_gemms()           # library GEMMs: tuned solution per shape (tunableop_gfx950.csv)
        from .optim import FusedAdam
        self.optimiser = FusedAdam(policy.parameters(), lr=self.cfg.lr, eps=self.cfg.eps)           # ppo.py:23 (+ the clip of :67 in its step)
        # gradients live in one persistent flat buffer with a fixed layout (dist.GradBucket): the all-reduce of a step is one
        # collective on that buffer, and zeroing the gradients is one memset
        self.bucket = cdist.GradBucket(policy.parameters(), assign_when_single_rank=True)
        dev = next(policy.parameters()).device
        self._sync = torch.cuda.synchronize if dev.type == "cuda" else (lambda: None)
        self.gen = torch.Generator(device=dev).manual_seed(seed + 17 * (1 + (torch.distributed.get_rank()
                                                                         if torch.distributed.is_initialized() else 0)))
        self.timings = {}

    def _autocast(self):
        if self.autocast_dtype is None:
            return torch.autocast(device_type="cuda", enabled=False)
        return torch.autocast(device_type="cuda", dtype=self.autocast_dtype)

    dedupe_boards = True          # run the tile encoder once per DISTINCT board: of a game's stored observations (value re-evaluation), of a minibatch
    dedupe_min_rows = 16384       # minibatches below this are launch-bound: the extra gather / scatter-add is not worth it

    @torch.no_grad()
    def board_runs(self, st):
        """For the stored observations [T+1, N]: which rows show a board (tile features: robber, buildings, relative owners) that
        differs from the game's previous stored observation - the only rows the tile encoder has to see - and, for every row, the
        position of its board in that list.  Under self-play with one active seat per game nine in ten consecutive observations
        repeat the board (trades, dice, development cards, end of turn do not touch it).  Computed once per update (the storage
        does not change between the epochs).  -> (first_rows int64 [U], board_of_row int64 [(T+1) N])"""
        from . import spec
        T1, N = st.obs_f.shape[0], st.obs_f.shape[1]
        o = spec.OBS_FLOAT_OFFSETS["tile_representations"]
        tiles = st.obs_f[:, :, o:o + 1140]
        new = torch.ones((T1, N), dtype=torch.bool, device=tiles.device)
        step = max(1, min(16, (256 << 20) // max(1, N * 1140)))   # (chunked by bytes: the comparison materialises a [t, N, 1140] mask, <= 256 MB)
        for t0 in range(1, T1, step):
            t1 = min(T1, t0 + step)
            new[t0:t1] = (tiles[t0:t1] != tiles[t0 - 1:t1 - 1]).any(-1)
        flat = new.reshape(-1)
        first_rows = flat.nonzero(as_tuple=True)[0]
        uid = torch.cumsum(flat.long(), 0) - 1                              # position in first_rows, valid where `new`
        uid = torch.where(flat, uid, torch.full_like(uid, -1)).reshape(T1, N)
        board_of_row = torch.cummax(uid, 0).values.reshape(-1)              # the last new board at or before t, per game
        return first_rows, board_of_row

    @torch.no_grad()
    def minibatch_boards(self, board_ids, perm, num_mini_batch, mbs):
        """The distinct boards of every minibatch of an epoch, for all minibatches at once (one sort, ONE host read of the 64
        counts - a `torch.unique` per step would drain the launch queue in the middle of every step).
        board_ids int64 [T N] (board_runs); perm: the epoch's permutation.  -> list of (boards int64 [U_k], inv int64 [mbs]):
        row j of minibatch k shows board boards[inv[j]]."""
        ids = board_ids[perm[:num_mini_batch * mbs]].view(num_mini_batch, mbs)
        srt, order = torch.sort(ids, dim=1)
        first = torch.ones_like(srt, dtype=torch.bool)
        first[:, 1:] = srt[:, 1:] != srt[:, :-1]
        pos = torch.cumsum(first.long(), 1) - 1                      # rank of every sorted entry's board among the distinct ones
        counts = (pos[:, -1] + 1).tolist()                           # the one host read
        uq = torch.zeros_like(srt)
        uq.scatter_(1, pos, srt)                                     # (duplicates write the same value)
        inv = torch.empty_like(pos)
        inv.scatter_(1, order, pos)
        # where each board's run starts among the rows sorted by board (for the per-board sums of the backward: nn_kernels.expand_rows)
        start = torch.full((num_mini_batch, mbs + 1), mbs, dtype=torch.int64, device=ids.device)
        start[:, :mbs].scatter_reduce_(1, pos, torch.arange(mbs, device=ids.device).expand(num_mini_batch, mbs), "amin", include_self=True)
        return [(uq[k, :counts[k]], inv[k], order[k], start[k, :counts[k] + 1]) for k in range(num_mini_batch)]

    def _gather_obs_parts(self, pol, f_all, idx, o):
        """the minibatch rows idx of the rollout's observation matrix f_all [rows, 1 787] without their tile features (columns o .. o + 1 139)
        -> policy.ObsParts when the net takes them (CUDA, the storage's dtype = the compute dtype), else one [len(idx), 1 787] matrix whose
        tile columns are not filled"""
        cast = (lambda x: x) if (self.autocast_dtype is not None and f_all.dtype == self.autocast_dtype) else (lambda x: x.float())
        parts_cls = getattr(sys.modules.get(type(pol).__module__), "ObsParts", None)
        n = idx.numel()
        if parts_cls is None or not f_all.is_cuda or f_all.dtype != torch.bfloat16 or self.autocast_dtype != torch.bfloat16 or o != 18 or f_all.shape[1] != 1787:
            fm = torch.empty((n, f_all.shape[1]), dtype=f_all.dtype, device=f_all.device)
            nn_kernels.gather_rows(f_all[:, :o], idx, out=fm[:, :o])
            nn_kernels.gather_rows(f_all[:, o + 1140:], idx, out=fm[:, o + 1140:])
            return cast(fm)
        buf = getattr(self, "_obs_parts_buf", None)
        if buf is None or buf[0].shape[0] != n or buf[0].device != f_all.device:
            # persistent: the opponents' pad column is zeroed once and never written again
            buf = self._obs_parts_buf = (torch.empty((n, 24), dtype=f_all.dtype, device=f_all.device), torch.empty((n, 152), dtype=f_all.dtype, device=f_all.device),
                                         torch.zeros((n, 3, 160), dtype=f_all.dtype, device=f_all.device))
        head, cur, oth = buf
        c0 = o + 1140
        nn_kernels.gather_rows(f_all[:, :o], idx, out=head[:, :o])
        nn_kernels.gather_rows(f_all[:, c0:c0 + 152], idx, out=cur)
        for j in range(3):
            nn_kernels.gather_rows(f_all[:, c0 + 152 + 159 * j:c0 + 152 + 159 * (j + 1)], idx, out=oth[:, j, :159])
        return parts_cls(head[:, :o], cur, oth.view(3 * n, 160))

    @torch.no_grad()
    def compute_values(self, st):
        """process_batch.py:108-132: V(obs) for all (T+1)*N observations, denormalised (RL/models/utils.py:20-21)."""
        T1, N = st.obs_f.shape[0], st.obs_f.shape[1]
        f = st.obs_f.reshape(T1 * N, -1); lists = st.lists.reshape(T1 * N, 5, -1); lens = st.lens.reshape(T1 * N, 5)
        out = torch.empty((T1 * N,), dtype=torch.float32, device=f.device)
        ch = self.cfg.value_chunk
        rec = getattr(self.policy, "include_lstm", False)
        cast = (lambda x: x) if (self.autocast_dtype is not None and f.dtype == self.autocast_dtype) else (lambda x: x.float())
        te_u = board_of_row = None
        vnet = self.policy
        if (self.autocast_dtype is not None and f.is_cuda and not rec and hasattr(self.policy, "inference_copy")
                and getattr(self.policy, "_inference_dtype", None) is None):
            # the value pass is inference: a copy of the net with its Linear weights already in the autocast dtype, refreshed once per
            # call (one copy per parameter) - under autocast the fp32 masters were re-cast in every chunk (1 700 launches per pass)
            if getattr(self, "_value_net", None) is None:
                self._value_net = self.policy.inference_copy(self.autocast_dtype)
            self._value_net.load_from(self.policy)
            vnet = self._value_net
        if self.dedupe_boards and not rec and f.is_cuda and hasattr(self.policy, "observation_module"):
            from . import spec
            key = (getattr(st, "token", None), getattr(st, "generation", None), T1, N)
            if getattr(self, "_runs_key", None) != key or getattr(st, "generation", None) is None or getattr(st, "token", None) is None:
                self._runs = self.board_runs(st)                # once per rollout: the storage does not change between the epochs
                self._runs_key = key
            first_rows, board_of_row = self._runs
            o = spec.OBS_FLOAT_OFFSETS["tile_representations"]
            tiles_all = f[:, o:o + 1140]
            for s in range(0, first_rows.numel(), ch):          # the tile encoder on the distinct boards only
                with self._autocast():
                    part = vnet.observation_module.tile_encoder(cast(nn_kernels.gather_rows(tiles_all, first_rows[s:s + ch])).reshape(-1, 19, 60))
                if te_u is None:
                    te_u = torch.empty((first_rows.numel(), part.shape[1]), dtype=part.dtype, device=part.device)
                te_u[s:s + ch] = part
        if rec:
            hid = st.hidden[:, :T1].reshape(2, T1 * N, -1); nt = st.masks[:T1].reshape(T1 * N)
        cast = (lambda x: x) if (self.autocast_dtype is not None and f.dtype == self.autocast_dtype) else (lambda x: x.float())
        for s in range(0, T1 * N, ch):
            with self._autocast():
                if rec:
                    v = self.policy.get_value(cast(f[s:s + ch]), lists[s:s + ch], lens[s:s + ch].long(),
                                              (hid[0, s:s + ch], hid[1, s:s + ch]), nt[s:s + ch])
                elif te_u is not None:
                    v = vnet.get_value(cast(f[s:s + ch]), lists[s:s + ch], lens[s:s + ch].long(), tile_features=nn_kernels.gather_rows(te_u, board_of_row[s:s + ch]))
                else:
                    v = vnet.get_value(cast(f[s:s + ch]), lists[s:s + ch], lens[s:s + ch].long())
            out[s:s + ch] = v[:, 0]
        return self.policy.denormalise(out).reshape(T1, N)

    def update(self, st):
        """-> (value_loss, action_loss, entropy_loss) averaged over the optimiser steps, as ppo.py:70-79."""
        cfg, pol = self.cfg, self.policy
        T, N = st.T, st.N
        dev = st.obs_f.device
        total = T * N
        mbs = total // cfg.num_mini_batch
        f_all = st.obs_f[:T].reshape(total, -1); lists_all = st.lists[:T].reshape(total, 5, -1); lens_all = st.lens[:T].reshape(total, 5)
        acts_all = st.actions.reshape(total, -1); amask_all = st.action_masks.reshape(total, -1)
        old_lp_all = st.action_log_probs.reshape(total)
        rewards = st.rewards[:T].contiguous(); masks = st.masks[:T + 1].contiguous()
        rec = getattr(pol, "include_lstm", False)
        nt_all = masks[:T].reshape(total)                 # masks_batch of generator_lstm (process_batch.py:249)
        cast = (lambda x: x) if (self.autocast_dtype is not None and f_all.dtype == self.autocast_dtype) else (lambda x: x.float())
        sums = torch.zeros(3, device=dev)                 # action loss, value loss, entropy (accumulated on device)
        # distinct boards (see board_runs): the encoder's share of a minibatch step shrinks with the boards a minibatch repeats
        dedupe = bool(self.dedupe_boards and not rec and dev.type == "cuda" and hasattr(pol, "observation_module") and mbs >= self.dedupe_min_rows)
        if dedupe:
            from . import spec
            key = (getattr(st, "token", None), getattr(st, "generation", None), T + 1, N)
            if getattr(self, "_runs_key", None) != key or getattr(st, "generation", None) is None or getattr(st, "token", None) is None:
                self._runs = self.board_runs(st)
                self._runs_key = key
            first_rows, board_of_row = self._runs
            board_ids = board_of_row[:total]
            o = spec.OBS_FLOAT_OFFSETS["tile_representations"]
            tiles_all = st.obs_f.reshape((T + 1) * N, -1)[:, o:o + 1140]
        timed_allreduce = dev.type == "cuda" and torch.distributed.is_initialized() and torch.distributed.get_world_size() > 1
        ar_events = []
        t_val = t_gae = t_opt = 0.0
        for _ in range(cfg.ppo_epoch):
            t0 = time.perf_counter()
            values = self.compute_values(st)                                               # ppo.py:31-32
            self._sync(); t1 = time.perf_counter()
            returns, adv = ppo_kernels.compute_gae(rewards, values, masks, cfg.gamma, cfg.gae_lambda)
            self._sync(); t2 = time.perf_counter()
            vpred = values[:T].reshape(total); ret = returns.reshape(total); advf = adv.reshape(total)
            if rec:
                seqs = lstm_minibatches(T, N, cfg.truncated_seq_len, cfg.num_mini_batch,
                                        torch.randperm(total // cfg.truncated_seq_len, generator=self.gen, device=dev))
                batches = [((t * N + g[None, :]).reshape(-1), (st.hidden[0, t[0], g], st.hidden[1, t[0], g])) for t, g in seqs]
            else:
                perm = torch.randperm(total, generator=self.gen, device=dev)               # SubsetRandomSampler
                batches = [(perm[mb * mbs:(mb + 1) * mbs], None) for mb in range(cfg.num_mini_batch)]   # BatchSampler(drop_last=True)
                if dedupe:                        # the tile encoder sees every distinct board of a minibatch once (forward and backward)
                    boards = self.minibatch_boards(board_ids, perm, cfg.num_mini_batch, mbs)
                ahm = getattr(pol, "action_head_module", None)
                groupings = None                  # the heads' row sets of every minibatch: one sort and one host read per epoch
                if ahm is not None and hasattr(ahm, "precompute_groupings") and dev.type == "cuda" and ahm.wants_grouping(mbs, acts_all):
                    groupings = ahm.precompute_groupings(acts_all, perm, cfg.num_mini_batch, mbs)
            # (the packed mask rows go to the net as they are: policy.PackedActionMasks expands what the heads read)
            pam = getattr(sys.modules.get(type(pol).__module__), "PackedActionMasks", None)
            amasks = (lambda i: pam(amask_all[i], st.unpack_action_masks)) if (pam is not None and dev.type == "cuda" and amask_all.dtype == torch.int32) \
                else (lambda i: st.unpack_action_masks(amask_all[i]))
            for bi, (idx, hidden) in enumerate(batches):
                with self._autocast():
                    if rec:
                        v, lp, ent, _ = pol.evaluate_actions(cast(f_all[idx]), lists_all[idx], lens_all[idx].long(),
                                                             amasks(idx), acts_all[idx],
                                                             hidden=hidden, nonterminal=nt_all[idx])     # ppo.py:48-50
                    elif dedupe:
                        uqk, invk, orderk, startk = boards[bi]
                        # the rows' observations WITHOUT their tile features (64 % of a row: they come per distinct board below), gathered
                        # straight into the pieces the observation module multiplies (policy.ObsParts: no slicing / padding copies later)
                        obs_in = self._gather_obs_parts(pol, f_all, idx, o)
                        v, lp, ent = pol.evaluate_actions(obs_in, nn_kernels.gather_rows(lists_all, idx), lens_all[idx].long(),
                                                          amasks(idx), acts_all[idx],
                                                          tile_dedupe=(cast(nn_kernels.gather_rows(tiles_all, first_rows[uqk])), invk, orderk, startk),
                                                          **({} if groupings is None else {"grouping": groupings[bi]}))
                    else:
                        v, lp, ent = pol.evaluate_actions(cast(f_all[idx]), lists_all[idx], lens_all[idx].long(),
                                                          amasks(idx), acts_all[idx],
                                                          **({} if groupings is None else {"grouping": groupings[bi]}))
                loss, parts = ppo_kernels.ppo_loss(lp.float(), v.float(), old_lp_all[idx], advf[idx], vpred[idx], ret[idx],
                                                   cfg.clip_param, cfg.value_loss_coef,
                                                   value_normaliser=(pol.VALUE_MEAN, pol.VALUE_STD))     # ppo.py:46-63
                self.bucket.zero()
                if dev.type == "cuda":
                    nn_kernels.grad_arena.begin_step(dev)                                  # the backward kernels' accumulators: one fill per step
                    nn_kernels.wgrad_queue.begin()                                         # tall-skinny weight gradients: grouped launches after the backward
                try:
                    (loss - ent * cfg.entropy_coef).backward()                             # ppo.py:66
                except BaseException:
                    nn_kernels.wgrad_queue.drop()      # (a flush here could raise in turn and hide the backward's own error)
                    raise
                nn_kernels.wgrad_queue.flush()
                if timed_allreduce:
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    self.bucket.allreduce()                                                # one RCCL all-reduce per step
                    e1.record()
                    ar_events.append((e0, e1))
                else:
                    self.bucket.allreduce()
                self.optimiser.step(cfg.max_grad_norm)                                     # ppo.py:67-68: clip_grad_norm_ + Adam, two launches (optim.FusedAdam)
                nn_kernels.grad_arena.end_step()
                nn_kernels.weight_images.refresh_all()                                     # every bf16 / transposed / packed image of the new weights: one launch
                sums += torch.stack((parts[0], parts[1], ent.detach().float()))
            self._sync(); t3 = time.perf_counter()
            t_val += t1 - t0; t_gae += t2 - t1; t_opt += t3 - t2
        n = cfg.ppo_epoch * len(batches)
        self.timings = {"values_s": t_val, "gae_s": t_gae, "minibatches_s": t_opt}
        if timed_allreduce:                               # device time inside the gradient all-reduces (part of minibatches_s)
            self._sync()
            self.timings["allreduce_s"] = sum(a.elapsed_time(b) for a, b in ar_events) * 1e-3
        al, vl, en = (sums / n).tolist()
        return vl * cfg.value_loss_coef, al, en * cfg.entropy_coef
