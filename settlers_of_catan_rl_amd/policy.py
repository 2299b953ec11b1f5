"""Policy / value network of the reference, restated for batched fixed-shape execution on MI355X (PyTorch-ROCm).

Architecture and semantics follow the reference (paths relative to the upstream repo):
  RL/models/build_agent_model.py:8-33,36-155   sizes, autoregressive map, type-conditional masks, log-prob masks
  RL/models/observation_module.py:47-61        tile encoder + player modules -> 987 -> 512 trunk
  RL/models/tile_encoder.py:67-91              60 -> 64, LN, ReLU, 2 pre-LN encoder layers (4 heads, FFN x2), 64 -> 25, LN, ReLU
  RL/models/player_modules.py:12-157           current / other player MLPs + masked dev-card attention, summed
  RL/models/multi_headed_attention.py:12-54    attention with a key mask
  RL/models/policy.py:52-111                   value MLP 512-256-128-1 with LayerNorm; act / evaluate_actions / get_value
  RL/models/action_heads_module.py:25-329      12 autoregressive heads, masked categoricals, recurrent resource heads
  RL/distributions.py:10-40                    logits = linear(x) + log(mask); entropy with p<=0 -> 1

Nothing here is copied: it is a from-scratch module tree whose *parameter names and shapes* equal the reference's
`state_dict` (the names are the checkpoint interface), so `load_state_dict(reference_sd, strict=False)` works and the
parity test can compare against the reference net with identical weights.  Differences by design:
  * inputs are the flat tensors the HIP encoder writes (obs float[B,1787], card-id lists int[B,5,25] + lengths, masks
    float[B,325]) instead of per-game Python dicts/lists; dev-card lists are fixed-pad 25 with a length mask;
  * actions are one int64 tensor [B,18] (heads 7/8 are 4-long sequences), see spec.ACTION_HEAD_SLICES;
  * the linears run under bf16 autocast on the GPU (MFMA via hipBLASLt); softmax/log-prob math stays fp32.
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import nn_kernels, spec

T_SETTLE, T_ROAD, T_CITY, T_BUYDEV, T_PLAYDEV, T_EXCHANGE, T_PROPOSE, T_RESPOND, T_ROBBER, T_ROLL, T_ENDTURN, T_STEAL, T_DISCARD = range(13)
C_YOP, C_MONO = 2, 4
MO = spec.MASK_OFFSETS
MASK_N = spec.MASK_WORDS


def _ortho_linear(i, o, gain=math.sqrt(2)):
    lin = nn.Linear(i, o)
    nn.init.orthogonal_(lin.weight, gain=gain)
    nn.init.zeros_(lin.bias)
    return lin


def _lin(x, w, b=None):
    """F.linear; tall-skinny shapes (huge row count, small widths) take the path whose weight gradient is the hand-written
    MFMA kernel (nn_kernels.linear) when training - as library GEMMs those gradients were 40 % of a step."""
    if nn_kernels.linear_supported(x, w):
        if torch.is_grad_enabled():
            return nn_kernels.linear(x, w, b)
        y = nn_kernels.linear_inference(x, w, b)          # inference: the forward kernel alone (10^6-row layers)
        if y is not None:
            return y
    return F.linear(x, w, b)


_PARTS_MIN_ROWS = 8192       # (inference: per-part accumulating products instead of concatenation + one product; tools/bench_parts_threshold.py:
                              #  the captured pass 2 065 -> 2 035 us at 65 536 rows, 1 233 -> 1 206 at 32 768, 908 -> 899 at 16 384)


def _lin_parts(parts, w, b):
    """Linear over the concatenation of `parts` along the last dim WITHOUT the concatenation: y = sum_k parts[k] @ W[:, cols_k].T
    + b (the reference concatenates - player_modules.py:66-69,109-111, observation_module.py:58-60 - and multiplies once: the
    same sums in another order).  Training on the GPU only: the concatenated widths (281, 306, 987) are not multiples of 8, so
    their weight gradients - tall-skinny products over 2 x 10^5..6 x 10^5 rows - went to the library at 0.6-0.8 ms each; the
    parts (256 / 128 / 384 / 25 wide) take the hand-written k_wgrad, and the concatenation's copy and its backward split go away."""
    ok, c = False, 0
    if torch.is_grad_enabled() and parts[0].is_cuda:
        for x in parts:
            ok = ok or nn_kernels.linear_supported(x, w[:, c:c + x.shape[-1]])
            c += x.shape[-1]
    if not ok:
        x0 = parts[0]
        rows = x0.numel() // x0.shape[-1]
        if (not torch.is_grad_enabled() and x0.is_cuda and x0.dtype == torch.bfloat16 and w.dtype == torch.bfloat16 and rows >= _PARTS_MIN_ROWS
                and all(p.dim() == 2 and p.dtype == torch.bfloat16 for p in parts)):
            # inference at 10^5..10^6 rows: one accumulating product per part instead of concatenation + one product - the rows of the
            # concatenation are 612 / 562 / 1 974 bytes and the copy kernel moves them at a third of the HBM rate (at 1 M rows:
            # 2.86 -> 2.13 ms for the final layer, 2.42 -> 1.25 ms for the opponents' module; tools/bench_cat_vs_parts.py)
            y, c = None, 0
            for x in parts:
                n = x.shape[-1]
                wt = w[:, c:c + n].t()
                y = (torch.addmm(b, x, wt) if b is not None else torch.mm(x, wt)) if y is None else y.addmm_(x, wt)
                c += n
            return y
        return F.linear(torch.cat(parts, -1), w, b)
    y, c = None, 0
    for k, x in enumerate(parts):
        n = x.shape[-1]
        t = _lin(x, w[:, c:c + n], b if k == 0 else None)
        y = t if y is None else y + t
        c += n
    return y


def _ln(ln, x, relu=False):
    """LayerNorm (+ optional ReLU): the fused HIP kernel for the small widths on the GPU, torch otherwise."""
    if nn_kernels.ln_supported(x, ln):
        return nn_kernels.small_layer_norm(x, ln, relu)
    y = ln(x)
    return F.relu(y) if relu else y


class _MHA(nn.Module):
    """qkv_nets.{0,1,2} + out_proj_net (reference multi_headed_attention.py:21-22)."""

    def __init__(self, dim, heads):
        super().__init__()
        self.heads, self.hd = heads, dim // heads
        self.qkv_nets = nn.ModuleList([nn.Linear(dim, dim) for _ in range(3)])
        self.out_proj_net = nn.Linear(dim, dim)

    def forward(self, x, lens=None, residual=None):
        """x [B, L, D]; lens [B] (keys >= len are masked, reference key mask) or None.  residual: added to the result (the
        encoder sub-layer's x + attention(norm(x)); fused into the out-projection on the GPU)."""
        y, added = self._forward(x, lens, residual)
        return y if residual is None or added else residual + y

    def _forward(self, x, lens, residual):
        B, L, D = x.shape
        w = torch.cat([n.weight for n in self.qkv_nets], 0)
        b = torch.cat([n.bias for n in self.qkv_nets], 0)
        qkv = _lin(x, w, b).view(B, L, 3, self.heads, self.hd)
        if x.is_cuda and nn_kernels.supported(L, self.heads, self.hd) and qkv.dtype in (torch.float32, torch.bfloat16):
            o = nn_kernels.small_attention(qkv, lens)             # fused HIP kernel (csrc/catan_nn.hip)
            if residual is not None and nn_kernels.fused_sublayer_supported(o, D):
                return nn_kernels.linear_residual(o, self.out_proj_net.weight, self.out_proj_net.bias, residual), True
            return _lin(o, self.out_proj_net.weight, self.out_proj_net.bias), False
        # reference formulation in plain torch ops (CPU parity tests, unsupported shapes)
        q, k, v = qkv.permute(2, 0, 3, 1, 4)
        scores = torch.matmul(q, k.transpose(-2, -1)) * (1.0 / math.sqrt(self.hd))
        if lens is not None:
            key_mask = torch.arange(L, device=x.device)[None, :] < lens[:, None]
            scores = scores.masked_fill(~key_mask[:, None, None, :], float("-inf"))
        o = torch.matmul(torch.softmax(scores.float(), -1).to(v.dtype), v)
        return self.out_proj_net(o.transpose(1, 2).reshape(B, L, D)), False


class _FFN(nn.Module):
    def __init__(self, dim, mult):
        super().__init__()
        self.linear1 = _ortho_linear(dim, mult * dim)
        self.linear2 = _ortho_linear(mult * dim, dim)

    def forward(self, x, residual=None):
        if residual is not None and nn_kernels.fused_sublayer_supported(x, self.linear1.in_features, self.linear1.out_features):
            return nn_kernels.ffn_residual(x, self.linear1.weight, self.linear1.bias, self.linear2.weight, self.linear2.bias, residual)
        y = _lin(F.relu(_lin(x, self.linear1.weight, self.linear1.bias)), self.linear2.weight, self.linear2.bias)
        return y if residual is None else residual + y


class _SubLayer(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.norm = nn.LayerNorm(dim)


class _EncoderLayer(nn.Module):
    def __init__(self, dim, heads):
        super().__init__()
        self.sublayers = nn.ModuleList([_SubLayer(dim), _SubLayer(dim)])
        self.multi_headed_attention = _MHA(dim, heads)
        self.pointwise_net = _FFN(dim, 2)

    def forward(self, x):
        if nn_kernels.pre_norm_supported(x, self.sublayers[0].norm):     # training on the GPU: x's gradient formed in the LayerNorm backward
            n, x = nn_kernels.pre_norm(x, self.sublayers[0].norm)
            x = self.multi_headed_attention(n, residual=x)
            n, x = nn_kernels.pre_norm(x, self.sublayers[1].norm)
            return self.pointwise_net(n, residual=x)
        x = self.multi_headed_attention(_ln(self.sublayers[0].norm, x), residual=x)
        return self.pointwise_net(_ln(self.sublayers[1].norm, x), residual=x)


class _TileEncoder(nn.Module):
    def __init__(self, in_dim=60, dim=64, heads=4, layers=2, out_dim=25):
        super().__init__()
        self.first_layer = _ortho_linear(in_dim, dim)
        self.encoder_layers = nn.ModuleList([_EncoderLayer(dim, heads) for _ in range(layers)])
        self.norm = nn.LayerNorm(out_dim)
        self.norm_2 = nn.LayerNorm(dim)
        self.out_proj = _ortho_linear(dim, out_dim)

    def forward(self, tiles, out_cols=None):
        """out_cols (training on the GPU only): the board rows zero-padded to that many columns (a multiple of 8: 16-byte rows)"""
        if nn_kernels.tile_encoder_supported(self, tiles):           # inference on the GPU: the whole encoder in one kernel
            return nn_kernels.tile_encoder_forward(self, tiles)
        if nn_kernels.tile_encoder_train_supported(self, tiles):     # training on the GPU: the same kernel, leaving what the backward reads
            return nn_kernels.tile_encoder_train(self, tiles, 19 * self.out_proj.out_features if out_cols is None else out_cols)
        assert out_cols is None
        w0 = self.first_layer.weight
        if tiles.is_cuda and tiles.shape[-1] % 8:                     # 60 features: zero-pad to 64 so the row kernels take the layer (as a
            pad = -tiles.shape[-1] % 8                                # strided 3-D F.linear it ran as a batched GEMM: 2.6 ms of a 55 ms step)
            tiles, w0 = F.pad(tiles, (0, pad)), F.pad(w0, (0, pad))
        x = _ln(self.norm_2, _lin(tiles, w0, self.first_layer.bias), relu=True)
        for layer in self.encoder_layers:
            x = layer(x)
        return _ln(self.norm, _lin(x, self.out_proj.weight, self.out_proj.bias), relu=True).reshape(tiles.shape[0], -1)   # relu(norm(.)).reshape == relu(norm(.).reshape)


def _card_summary(ids, lens, embedding, mha, norm):
    """Masked attention over a padded card-id list, zero the padding, sum (player_modules.py:55-69) - computed per card
    CLASS instead of per token.  A list holds at most 6 distinct ids, and everything the reference computes per token
    (embedding, Q/K/V, attention output, out-projection, LayerNorm) depends only on the token's id and on how many valid
    tokens of each id the list has: softmax over keys j of s(id_i, id_j) equals softmax over classes b of
    s(id_i, b) + log(count_b), so the whole module is a function of the six counts.  Same value as the per-token form
    (it merges identical terms of the same sums).  On the GPU it is one fused kernel per list (nn_kernels.card_summary) fed
    with the 4 x 6 x 6 score table and the six value vectors; elsewhere the same algebra in torch ops."""
    B, L = ids.shape
    V, H, hd = embedding.num_embeddings, mha.heads, mha.hd
    if not torch.is_grad_enabled() and nn_kernels.card_summary_supported(ids, V, H, hd, H * hd):
        return nn_kernels.card_summary_lookup(ids, lens, embedding, mha, norm)       # inference: a table of the 4 860 count patterns
    with torch.autocast(device_type=ids.device.type, enabled=False):          # a few hundred flops per row: keep them fp32
        w = torch.cat([n.weight for n in mha.qkv_nets], 0).float()
        b = torch.cat([n.bias for n in mha.qkv_nets], 0).float()
        qkv = F.linear(embedding.weight.float(), w, b).view(V, 3, H, hd)      # Q / K / V of each of the six ids
        s_ab = torch.einsum("ahd,bhd->hab", qkv[:, 0], qkv[:, 1]) * (1.0 / math.sqrt(hd))
        if nn_kernels.card_summary_supported(ids, V, H, hd, H * hd):           # GPU: one fused kernel, 26 B in / 64 B out per list
            params = torch.cat((s_ab.reshape(-1), qkv[:, 2].reshape(-1), mha.out_proj_net.weight.float().reshape(-1),
                                mha.out_proj_net.bias.float(), norm.weight.float(), norm.bias.float()))
            return nn_kernels.card_summary(ids, lens, params, norm.eps)
        valid = (torch.arange(L, device=ids.device)[None, :] < lens[:, None]).float()
        cnt = torch.zeros((B, V), dtype=torch.float32, device=ids.device).scatter_add_(1, ids.long(), valid)     # valid tokens per id
        p = torch.softmax(s_ab[None] + torch.log(cnt)[:, None, None, :], -1)  # [B, H, 6, 6]; absent classes: log 0 = -inf
        attn = torch.einsum("rhab,bhd->rahd", p, qkv[:, 2]).reshape(B, V, H * hd)
        rep = F.layer_norm(F.linear(attn, mha.out_proj_net.weight.float(), mha.out_proj_net.bias.float()), norm.normalized_shape,
                           norm.weight.float(), norm.bias.float(), norm.eps)
        return (rep * cnt[..., None]).sum(1)


class _CurrentPlayer(nn.Module):
    def __init__(self, in_dim=152, card_dim=16, proj=25):
        super().__init__()
        self.main_input_layer_1 = _ortho_linear(in_dim, 256)
        self.norm = nn.LayerNorm(card_dim)
        self.norm_1 = nn.LayerNorm(256)
        self.norm_2 = nn.LayerNorm(proj)
        self.norm_3 = nn.LayerNorm(proj)
        self.norm_4 = nn.LayerNorm(128)
        self.proj_hidden_dev_card = _ortho_linear(card_dim, proj)
        self.proj_played_dev_card = _ortho_linear(card_dim, proj)
        self.final_linear_layer = _ortho_linear(2 * proj + 256, 128)

    def forward(self, main, hid, hid_len, played, played_len, emb, hid_mha, played_mha):
        h = _ln(self.norm_2, _lin(_card_summary(hid, hid_len, emb, hid_mha, self.norm), self.proj_hidden_dev_card.weight, self.proj_hidden_dev_card.bias), relu=True)
        p = _ln(self.norm_3, _lin(_card_summary(played, played_len, emb, played_mha, self.norm), self.proj_played_dev_card.weight, self.proj_played_dev_card.bias), relu=True)
        m = _ln(self.norm_1, _lin(main, self.main_input_layer_1.weight, self.main_input_layer_1.bias), relu=True)
        return _ln(self.norm_4, _lin_parts((m, p, h), self.final_linear_layer.weight, self.final_linear_layer.bias), relu=True)


class _OtherPlayers(nn.Module):
    def __init__(self, in_dim=159, card_dim=16, proj=25):
        super().__init__()
        self.main_input_layer_1 = _ortho_linear(in_dim, 256)
        self.proj_played_dev_card = _ortho_linear(card_dim, proj)
        self.final_linear_layer = _ortho_linear(proj + 256, 128)
        self.norm = nn.LayerNorm(card_dim)
        self.norm_1 = nn.LayerNorm(256)
        self.norm_2 = nn.LayerNorm(proj)
        self.norm_3 = nn.LayerNorm(128)

    def forward(self, main, played, played_len, emb, played_mha):
        p = _ln(self.norm_2, _lin(_card_summary(played, played_len, emb, played_mha, self.norm), self.proj_played_dev_card.weight, self.proj_played_dev_card.bias), relu=True)
        w1 = self.main_input_layer_1.weight
        if main.shape[-1] > w1.shape[1] and not (nn_kernels.linear_supported(main, w1) and torch.is_grad_enabled()):
            main = main[:, :w1.shape[1]]                       # (rows zero-padded to whole 16-byte pieces - ObsParts.others - are for nn_kernels.linear)
        m = _ln(self.norm_1, _lin(main, w1, self.main_input_layer_1.bias), relu=True)
        return _ln(self.norm_3, _lin_parts((m, p), self.final_linear_layer.weight, self.final_linear_layer.bias), relu=True)


class _ObservationModule(nn.Module):
    def __init__(self):
        super().__init__()
        self.dev_card_embedding = nn.Embedding(6, 16)
        self.hidden_card_mha = _MHA(16, 4)
        self.played_card_mha = _MHA(16, 4)
        self.tile_encoder = _TileEncoder()
        self.current_player_module = _CurrentPlayer()
        self.other_players_module = _OtherPlayers()
        self.final_layer = _ortho_linear(19 * 25 + 4 * 128, 512)
        self.norm = nn.LayerNorm(512)

    def forward(self, obs_f, lists, lens, tile_features=None, tile_dedupe=None):
        """tile_features [B, 475]: the tile encoder's output for these rows, computed by the caller (the value re-evaluation of a
        PPO update encodes every DISTINCT board once: nine in ten consecutive observations of a game show the same board).
        tile_dedupe = (tiles_u [U, 19 * 60], inv int64 [B]): the DISTINCT boards of this batch and, per row, which of them it
        shows - the encoder runs on the U boards, its output is spread to the rows by `inv` and the rows' gradients are summed
        per board on the way back (the same function of the same inputs: identical boards give identical encodings)."""
        o = spec.OBS_FLOAT_OFFSETS
        B = obs_f.shape[0]
        parts = obs_f if isinstance(obs_f, ObsParts) else None
        if parts is not None:
            if tile_dedupe is None and tile_features is None:
                raise ValueError("ObsParts carries no tile features: pass tile_dedupe or tile_features")
            tiles, cur = None, parts.cur
        else:
            tiles = obs_f[:, o["tile_representations"]:o["tile_representations"] + 1140].reshape(B, 19, 60)
            cur = obs_f[:, o["current_player_main"]:o["current_player_main"] + 152]
        br = _OBS_BRANCHES.fork(cur)         # inference: the three independent parts on forked streams (see _Branches)
        with br.on(1):
            te_pad = 0
            tiles_u = None if tile_dedupe is None else tile_dedupe[0].reshape(-1, 19, 60)
            if tile_dedupe is not None and len(tile_dedupe) == 4 and nn_kernels.tile_encoder_train_supported(self.tile_encoder, tiles_u):
                # per distinct board, its 475 columns padded to 480 (960-byte rows), spread to the rows showing the board by
                # nn_kernels.expand_rows, whose backward sums the rows' gradients per board
                te = nn_kernels.expand_rows(self.tile_encoder(tiles_u, out_cols=480), *tile_dedupe[1:])
                te_pad = te.shape[1] - 475
            elif tile_dedupe is not None:
                te = self.tile_encoder(tile_dedupe[0].reshape(-1, 19, 60))[tile_dedupe[1]]    # (indexing: its backward sums per board by sorting - index_select's uses bf16 atomics, twice as slow)
            else:
                te = self.tile_encoder(tiles) if tile_features is None else tile_features
            te = br.keep(te)
        with br.on(2):
            cp = br.keep(self.current_player_module(cur, lists[:, 1], lens[:, 1], lists[:, 0], lens[:, 0], self.dev_card_embedding,
                                                    self.hidden_card_mha, self.played_card_mha))
        # the three opponents share one module: run them as one batch of 3B rows
        k0 = o["next_player_main"]
        others = parts.others if parts is not None else obs_f[:, k0:k0 + 3 * 159].reshape(B * 3, 159)
        op = self.other_players_module(others, lists[:, 2:5].reshape(B * 3, -1), lens[:, 2:5].reshape(B * 3),
                                       self.dev_card_embedding, self.played_card_mha)
        br.join()
        if te_pad:
            # (the concatenation is 992 wide - the reference's 987 columns and the 5 zero ones: one library product, as _lin_parts
            # decides for these widths; the pad columns meet zero weights)
            w = self.final_layer.weight
            wp = torch.cat((w[:, :475], w.new_zeros((w.shape[0], te_pad)), w[:, 475:]), 1)
            xcat = nn_kernels.concat_rows((te, cp, op.reshape(B, 3 * 128)))
            if xcat.is_cuda and torch.is_grad_enabled() and xcat.dtype == torch.bfloat16 and nn_kernels.wgrad_big_supported(B, wp.shape[1], wp.shape[0]):
                return _ln(self.norm, nn_kernels.linear_big(xcat, wp, self.final_layer.bias), relu=True)     # weight gradient: k_wgrad_big
            return _ln(self.norm, F.linear(xcat, wp, self.final_layer.bias), relu=True)
        return _ln(self.norm, _lin_parts((te, cp, op.reshape(B, 3 * 128)), self.final_layer.weight, self.final_layer.bias), relu=True)


class _Dist(nn.Module):
    def __init__(self, i, o):
        super().__init__()
        self.linear = _ortho_linear(i, o, gain=0.01)


class _Head(nn.Module):
    """mlp_1 -> LayerNorm -> ReLU -> mlp_2 -> distribution.linear (action_heads_module.py:202-228)."""

    def __init__(self, in_dim, out_dim, custom_in=0, custom_out=0):
        super().__init__()
        if custom_in:
            self.custom_mlp = nn.Linear(custom_in, custom_out)
            self.custom_norm = nn.LayerNorm(custom_out)
        self.mlp_1 = nn.Linear(in_dim + custom_out, 128)
        self.mlp_2 = nn.Linear(128, 128)
        self.norm = nn.LayerNorm(128)
        self.distribution = _Dist(128, out_dim)

    def logits(self, pre, extra=None, custom=None):
        """pre [B,128]: mlp_1 applied to the trunk columns of its input, bias included - `_ActionHeads` computes it for all
        twelve heads in ONE GEMM (the trunk is the same for every head and for every step of the recurrent heads; as
        separate layers that was 18 GEMMs over a [B, 512+] input plus a concatenation of the trunk with a few conditioning
        columns each).  extra [B,e] / custom: the conditioning columns that follow the trunk in mlp_1's input."""
        parts = [] if extra is None else [extra]
        if custom is not None:
            parts.append(_ln(self.custom_norm, _lin(custom, self.custom_mlp.weight, self.custom_mlp.bias), relu=True))
        if parts:
            e = parts[0] if len(parts) == 1 else torch.cat(parts, -1)
            pre = pre + _lin(e.to(pre.dtype), self.mlp_1.weight[:, self.mlp_1.in_features - e.shape[-1]:])     # (tall-skinny weight gradient: 2..12 columns x 10^4..10^5 rows)
        # the 128 -> 128 and 128 -> (2..73) layers: their weight gradients are tall-skinny products over the batch rows (_lin)
        h = _lin(_ln(self.norm, pre, relu=True), self.mlp_2.weight, self.mlp_2.bias)
        return _lin(h, self.distribution.linear.weight, self.distribution.linear.bias).float()


class ObsParts(object):
    """The observation rows of a learner minibatch ALREADY split into the pieces the observation module multiplies (the gather that
    builds the minibatch writes them there): `head` [B, 18] (proposed trade | current resources), `cur` [B, 152], `others` [3 B, 160]
    (the three opponents' 159 features as rows, zero-padded to a whole number of 16-byte pieces).  As one [B, 1 787] row matrix every
    piece was sliced out again and copied - and the opponents' rows padded - before its product: 0.23 ms of a minibatch step.  The tile
    features come through `tile_dedupe` (the learner encodes every distinct board once)."""

    def __init__(self, head, cur, others):
        self.head, self.cur, self.others = head, cur, others
        self.shape, self.device, self.is_cuda, self.dtype = (cur.shape[0], spec.OBS_FLOATS), cur.device, cur.is_cuda, cur.dtype


class PackedActionMasks(object):
    """The action masks of a batch as the env's packed rows (int32 [B, 11]: bit i of the flat 325-entry mask = word i >> 5, bit i & 31),
    expanded on demand.  The learner's compact evaluation needs the 13 type bits of every row and the full rows of the ~170 000 rows
    some head runs on: expanding all 204 800 rows to fp32 (266 MB) and gathering 1 300-byte rows out of that (another 440 MB moved) was
    0.24 ms of a minibatch step; gathering the 44-byte packed rows and expanding those is 0.07."""

    def __init__(self, packed, unpack):
        self.packed, self.unpack, self._dense = packed, unpack, None
        self.shape, self.device, self.is_cuda = (packed.shape[0], MASK_N), packed.device, packed.is_cuda

    def float(self):
        return self

    def dense(self):
        if self._dense is None:
            self._dense = self.unpack(self.packed)
        return self._dense

    def rows(self, idx):
        return self.unpack(self.packed[idx])

    def type_mask(self):
        w = self.packed[:, 0]
        return ((w[:, None] >> torch.arange(13, device=w.device, dtype=w.dtype)) & 1).float()


def _masked_logp(logits, mask):
    """log-softmax of logits + log(mask) (distributions.py:35-38)."""
    return F.log_softmax(logits + torch.log(mask), dim=-1)


def _entropy(logp):
    p = logp.exp()
    return -(p * torch.where(p > 0, logp, torch.zeros_like(logp))).sum(-1)      # p <= 0 -> contributes 0 (distributions.py:18-20)


def _categorical(logits, mask, given, deterministic, generator):
    """-> (action [B], log-prob of the action [B], entropy [B]) of the masked categorical (distributions.py:10-40): one HIP
    kernel on the GPU (nn_kernels.masked_categorical), the torch formulation otherwise."""
    if nn_kernels.categorical_supported(logits):
        return nn_kernels.masked_categorical(logits, mask, given, deterministic, generator)
    lp = _masked_logp(logits, mask)
    a = _choose(lp, given, deterministic, generator)
    return a, lp.gather(-1, a[:, None]).squeeze(-1), _entropy(lp)


def _choose(logp, given, deterministic, generator):
    if given is not None:
        return given
    if deterministic:
        return logp.argmax(-1)
    return torch.multinomial(logp.exp(), 1, generator=generator).squeeze(-1)


class _Branches(object):
    """Independent branches of an INFERENCE pass on side streams.  A policy pass at rollout width is a few hundred kernels of
    5-15 us whose cost is their launch-to-launch latency, not their work; enqueued on forked streams (and captured that way
    into the hipGraph of `GraphedAct`) independent chains - the tile encoder beside the player modules, the action heads that
    depend on the sampled type only - overlap.  Program order (and with it the order of the random draws) is unchanged; only
    the stream a branch is enqueued on differs.  Off under autograd and on the CPU."""
    N_SIDE = 3
    enabled = True
    # NOT inside a stream capture (round 5).  A hipGraph with parallel branches is launched by the HIP runtime on internal streams of
    # its own, chosen by a search that skips the streams sharing the launch stream's hardware queue - and that search has no bound:
    # which queue a new stream lands on depends on every stream the process has created and destroyed before (each env handle creates
    # three), and when too few of the graph's streams are on another queue hipGraphLaunch reads past the end of the vector and the
    # process dies with SIGSEGV inside libamdhip64 (tools/rollout_schedules.py's fourth env + collector in round 4; reproduced with a
    # native backtrace, gone with single-chain graphs or DEBUG_HIP_FORCE_GRAPH_QUEUES=1: profiles/r05_graph_launch_segv.txt).  A
    # captured policy pass is therefore ONE chain; the branches cost 0.12 s of a 2.3 s rollout at 65 536 games (same file).
    # The work-around is GATED on the runtime (round 6): branched graphs come back on a HIP runtime listed in GOOD_RUNTIMES (hipRuntimeGetVersion
    # values on which tools/repro_four_collectors.py with CATAN_GRAPH_BRANCHES=1 completes its six sets - none is known yet: the ROCm 7.2.0
    # runtime of this image, 70253xxx, has the defect) or with CATAN_GRAPH_BRANCHES=1; CATAN_GRAPH_BRANCHES=0 keeps single chains everywhere.
    GOOD_RUNTIMES = frozenset()
    _in_graphs = None

    @classmethod
    def graphs_allowed(cls):
        if cls._in_graphs is None:
            import os
            v = os.environ.get("CATAN_GRAPH_BRANCHES")
            if v in ("0", "1"):
                cls._in_graphs = v == "1"
            else:
                try:
                    from . import _lib
                    cls._in_graphs = int(_lib.lib().catan_hip_runtime_version()) in cls.GOOD_RUNTIMES
                except Exception:
                    cls._in_graphs = False
        return cls._in_graphs

    @property
    def in_graphs(self):
        return type(self).graphs_allowed()

    def __init__(self):
        self.side = None
        self.active = False

    def fork(self, ref):
        self.active = bool(self.enabled and ref.is_cuda and not torch.is_grad_enabled()
                           and (not torch.cuda.is_current_stream_capturing() or self.in_graphs))
        if not self.active:
            return self
        if self.side is None or self.side[0].device != ref.device:
            self.side = [torch.cuda.Stream(device=ref.device) for _ in range(self.N_SIDE)]
        self.main = torch.cuda.current_stream(ref.device)
        for st in self.side:
            st.wait_stream(self.main)
        self.made = []
        return self

    def on(self, k):
        """context: branch k (0 = stay on the main stream)"""
        import contextlib
        if not self.active or k % (self.N_SIDE + 1) == 0:
            return contextlib.nullcontext()
        return torch.cuda.stream(self.side[k % (self.N_SIDE + 1) - 1])

    def keep(self, *tensors):
        """tensors made on a side stream that the main stream reads after join()"""
        if self.active:
            self.made.extend(t for t in tensors if torch.is_tensor(t))
        return tensors[0] if len(tensors) == 1 else tensors

    def join(self):
        if not self.active:
            return
        for st in self.side:
            self.main.wait_stream(st)
        for t in self.made:
            t.record_stream(self.main)
        self.made = []
        self.active = False


_HEAD_BRANCHES, _OBS_BRANCHES, _VALUE_BRANCHES = _Branches(), _Branches(), _Branches()


def _cast_w(w, dtype):
    """a parameter (view) in the compute dtype: the registered bf16 image (nn_kernels.weight_images) when there is one"""
    return nn_kernels.bf16_of(w) if (dtype == torch.bfloat16 and w.is_cuda) else w.to(dtype)


class _SegmentedTrunk(torch.autograd.Function):
    """out_k = x[a_k:b_k] @ w_k.T + bias_k for a list of row segments of ONE gathered trunk tensor x (the heads of
    `_evaluate_compact`: every head sees its own rows).  As separate slices + F.linear the backward of every slice
    materialises a zero tensor of the whole of x, copies its rows in and adds it to the gradient of x (ten times 170 000 x 512
    per minibatch); here the backward writes every segment of dx once.  The segments must cover x; they may repeat (two heads
    on the same rows).  Same arithmetic as F.linear under the caller's autocast dtype."""

    @staticmethod
    def forward(ctx, x, segs, *wb):
        n = len(segs)
        ws, bs = wb[:n], wb[n:]
        outs = []
        for (a, b), w, bias in zip(segs, ws, bs):
            outs.append(F.linear(x[a:b], _cast_w(w, x.dtype), _cast_w(bias, x.dtype)))
        ctx.segs = segs
        ctx.save_for_backward(x, *ws)
        return tuple(outs)

    @staticmethod
    def backward(ctx, *douts):
        x, ws = ctx.saved_tensors[0], ctx.saved_tensors[1:]
        dx = torch.empty_like(x)
        seen = set()
        n = len(ctx.segs)
        dws, dbs = [None] * n, [None] * n
        grouped = []                                            # (segment index, x rows, dy): their weight gradients in shared launches
        for k, ((a, b), w, g) in enumerate(zip(ctx.segs, ws, douts)):
            if g is None:
                continue
            g = g.to(x.dtype)
            contrib = g @ _cast_w(w, x.dtype)
            if (a, b) in seen:
                dx[a:b] += contrib
            else:
                dx[a:b] = contrib
                seen.add((a, b))
            if x.is_cuda and x.dtype == torch.bfloat16 and b - a >= 4096 and nn_kernels.wgrad_supported(b - a, x.shape[1], g.shape[1]):
                grouped.append((k, x[a:b], g))                  # MFMA weight-gradient kernel (column slices of the 512-wide trunk)
            else:
                dws[k] = (g.t() @ x[a:b]).to(w.dtype)
                dbs[k] = g.float().sum(0)
        if grouped:
            if GROUPED_WGRAD:
                for (k, _, _), (dw, db) in zip(grouped, nn_kernels.wgrad_grouped([(xs, g) for _, xs, g in grouped])):
                    dws[k], dbs[k] = dw.to(ws[k].dtype), db
            else:
                for k, xs, g in grouped:
                    dw, db = nn_kernels.wgrad(xs, g)
                    dws[k], dbs[k] = dw.to(ws[k].dtype), db
        for (a, b) in ctx.segs:
            if (a, b) not in seen:
                dx[a:b] = 0
                seen.add((a, b))
        return (dx, None) + tuple(dws) + tuple(dbs)


GROUPED_WGRAD = True        # (A/B switch, tools/ab_step_switches.py)
RECURRENT_BATCHED = True    # (A/B switch) the trade heads' four steps as one pass when the actions are given


class _ActionHeads(nn.Module):
    def __init__(self, D=512):
        super().__init__()
        self.D = D
        self.action_heads = nn.ModuleList([
            _Head(D, 13), _Head(D + 2, 54), _Head(D, 73), _Head(D, 19), _Head(D, 5),
            _Head(D, 2, custom_in=12, custom_out=32), _Head(D + 2, 3), _Head(D + 6, 6), _Head(D + 6 + 6, 6),
            _Head(D + 4, 5), _Head(D + 4 + 5, 5), _Head(D, 5)])

    # Evaluating GIVEN actions (the PPO update): a head contributes to the joint log-prob and to the entropy term only on
    # the rows whose action type uses it (`log_prob_masks`, build_agent_model.py:132-147: its log-prob and entropy are
    # multiplied by 0 everywhere else), so every head but the type head runs on just those rows - a few per cent of the
    # batch each.  Same value, same gradient (the skipped rows' terms are exact zeros); 27 -> 8 ms at 204 800 rows.
    compact_evaluate = True
    compact_min_rows = 49152       # below this a minibatch step is launch-bound and the extra gathers (and the host read) cost more than they save

    def wants_grouping(self, rows, actions):
        return actions is not None and self.compact_evaluate and rows >= self.compact_min_rows

    def start_grouping(self, actions):
        """The row sets of `_evaluate_compact` depend on the ACTIONS only: sort by (type, card of a played development card)
        and count.  Called before the observation module is launched, with the counts on their way to pinned host memory
        behind an event - by the time the heads need them on the host the copy is long done, so the host read no longer
        drains the launch queue in the middle of a step (it cost ~7 ms of idle GPU per 54 ms minibatch step)."""
        typ, card = actions[:, 0], actions[:, 4]
        key = typ * 8 + torch.where(typ == T_PLAYDEV, card.clamp(0, 7), torch.zeros_like(card))
        perm = torch.argsort(key, stable=True)
        ends = torch.cumsum(torch.bincount(key, minlength=13 * 8), 0)
        if not actions.is_cuda:
            return perm, ends, None
        pin = getattr(self, "_ends_pinned", None)
        if pin is None or pin.numel() != ends.numel():
            pin = self._ends_pinned = torch.empty(ends.numel(), dtype=ends.dtype, pin_memory=True)
        pin.copy_(ends, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        return perm, pin, ev

    @torch.no_grad()
    def precompute_groupings(self, actions_all, perm, num_mini_batch, mbs):
        """start_grouping for every minibatch of an epoch at once: one batched sort, ONE host read of the 64 x 104 counts.  A read per
        step - even behind an event - holds the host at the start of every step until the device has finished the previous one, and
        the ~1 500 launches of a step then go out with the device waiting for each of them (the step is host-paced from there on).
        actions_all [rows, 18]; perm: the epoch's permutation.  -> list of (perm_k, ends_k (host list), None)"""
        rows = perm[:num_mini_batch * mbs]                  # (only the two columns the key is made of are gathered, not all 18)
        typ, card = actions_all[:, 0][rows].view(num_mini_batch, mbs), actions_all[:, 4][rows].view(num_mini_batch, mbs)
        key = typ * 8 + torch.where(typ == T_PLAYDEV, card.clamp(0, 7), torch.zeros_like(card))
        order = torch.argsort(key, dim=1, stable=True)
        counts = torch.zeros((num_mini_batch, 13 * 8), dtype=torch.int64, device=typ.device)
        counts.scatter_add_(1, key, torch.ones_like(key))
        ends = torch.cumsum(counts, 1).tolist()
        return [(order[k], ends[k], None) for k in range(num_mini_batch)]

    def _evaluate_compact(self, main, m, cur_res, trade, actions, grouping=None, value_fn=None):
        B, dev, H, D = main.shape[0], main.device, self.action_heads, self.D
        typ, card = actions[:, 0], actions[:, 4]
        pre_of = lambda i, x: _lin(x, H[i].mlp_1.weight[:, :D], H[i].mlp_1.bias)
        packed = isinstance(m, PackedActionMasks)
        # packed masks on the GPU: the heads read their mask as BITS of the env's rows (nn_kernels.masked_categorical_bits) - no float mask
        # matrix, no mask gather, no concatenation of the type-dependent mask rows
        bits = packed and main.is_cuda and nn_kernels.CATEGORICAL_BITS
        if bits:
            type_head = lambda x: nn_kernels.masked_categorical_bits(H[0].logits(pre_of(0, x)), m.packed, None, [(B, MO[0], None)], typ)
        else:
            type_head = lambda x: _categorical(H[0].logits(pre_of(0, x)), m.type_mask() if packed else m[:, MO[0]:MO[0] + 13], typ, False, None)
        # one sort by (type, card of a played development card) and one host read give every head's rows
        perm, ends, ev = grouping if grouping is not None else self.start_grouping(actions)
        if ev is not None:
            ev.synchronize()
        if torch.is_tensor(ends):
            ends = ends.tolist()
        rng = lambda k: perm[(ends[k - 1] if k else 0):ends[k]]
        of_type = lambda t: perm[(ends[8 * t - 1] if t else 0):ends[8 * t + 7]]
        rs, rc, rp, rst = of_type(T_SETTLE), of_type(T_CITY), of_type(T_PROPOSE), of_type(T_STEAL)
        rex, ryop, rmono = of_type(T_EXCHANGE), rng(8 * T_PLAYDEV + C_YOP), rng(8 * T_PLAYDEV + C_MONO)
        sets = {1: torch.cat((rs, rc)), 2: of_type(T_ROAD), 3: of_type(T_ROBBER), 4: of_type(T_PLAYDEV), 5: of_type(T_RESPOND),
                6: torch.cat((rp, rst)), 7: rp, 9: torch.cat((rex, ryop, rmono)), 10: torch.cat((rex, ryop)), 11: of_type(T_DISCARD)}
        # ONE gather of the trunk rows (and of the mask / action rows) for all heads: its backward is one scatter-add
        order = [i for i in sets if sets[i].numel()]
        if not order:
            self.last_value = value_fn(main) if value_fn is not None else None
            _, logp0, e0 = type_head(main)
            return actions.clone(), logp0, e0.sum() / B
        all_rows = torch.cat([sets[i] for i in order])
        # the same lists as ranges of perm, (a, b, where the range starts in all_rows): the gather's backward sums a row's <= 3 gradients
        # in one pass (nn_kernels.gather_ranges) instead of an index_put that sorts the indices again in every step
        t_r = lambda t: ((ends[8 * t - 1] if t else 0), ends[8 * t + 7])
        k_r = lambda k: ((ends[k - 1] if k else 0), ends[k])
        sets_r = {1: (t_r(T_SETTLE), t_r(T_CITY)), 2: (t_r(T_ROAD),), 3: (t_r(T_ROBBER),), 4: (t_r(T_PLAYDEV),), 5: (t_r(T_RESPOND),),
                  6: (t_r(T_PROPOSE), t_r(T_STEAL)), 7: (t_r(T_PROPOSE),), 9: (t_r(T_EXCHANGE), k_r(8 * T_PLAYDEV + C_YOP), k_r(8 * T_PLAYDEV + C_MONO)),
                  10: (t_r(T_EXCHANGE), k_r(8 * T_PLAYDEV + C_YOP)), 11: (t_r(T_DISCARD),)}
        pieces, off = [], 0
        for i in order:
            for a_, b_ in sets_r[i]:
                if b_ > a_:
                    pieces.append((a_, b_, off)); off += b_ - a_
        # `main` has three consumers - the type head, the value head (value_fn), this gather: with all three behind ONE autograd node its
        # gradient is formed in the gather's backward pass (nn_kernels.fanout_gather_ranges) instead of two more full-size adds
        m_type = m_val = main
        if off != all_rows.numel():
            mg = main[all_rows]
        elif value_fn is not None:
            m_type, m_val, mg = nn_kernels.fanout_gather_ranges(main, perm, all_rows, pieces)
        else:
            mg = nn_kernels.gather_ranges(main, perm, all_rows, pieces)
        self.last_value = value_fn(m_val) if value_fn is not None else None
        _, logp0, e0 = type_head(m_type)
        mm_g, ag = (None if bits else m.rows(all_rows) if packed else m[all_rows]), actions[all_rows]
        at, off = {}, 0
        for i in order:
            at[i] = slice(off, off + sets[i].numel()); off += sets[i].numel()
        lps, ents = [], []                                   # (slice, log-probs) per head; entropies
        # every head's trunk projection on its own rows, as ONE autograd node over the gathered rows (head 8 shares head 7's rows)
        jobs = [(j, at[i]) for i in order for j in ((7, 8) if i == 7 else (i,))]
        pres = _SegmentedTrunk.apply(mg, tuple((sl.start, sl.stop) for _, sl in jobs), *[H[j].mlp_1.weight[:, :D] for j, _ in jobs],
                                     *[H[j].mlp_1.bias for j, _ in jobs])
        pre = {j: p for (j, _), p in zip(jobs, pres)}

        def run(i, sl, mask, col, extra=None, custom=None):
            """mask: a float [rows, K] matrix - or, with `bits`, the (row count, bit offset, AND offset or None) segments of the head's rows"""
            logits = H[i].logits(pre[i], extra, custom)
            if bits:
                _, lp, ent = nn_kernels.masked_categorical_bits(logits, m.packed, all_rows[sl], mask, ag[sl, col])
            else:
                _, lp, ent = _categorical(logits, mask, ag[sl, col], False, None)
            lps.append((sl, lp)); ents.append(ent)

        def two_hot(n_first, n):
            x = torch.zeros(n, 2, device=dev); x[:n_first, 0] = 1.0; x[n_first:, 1] = 1.0
            return x

        if 1 in at:        # corner (settlement / city), its mask row picked by the type (build_agent_model.py:113-115)
            sl, n1 = at[1], rs.numel()
            nr = sl.stop - sl.start
            if bits:
                mask = [(n1, MO[1], None), (nr - n1, MO[1] + 54, None)]
            else:
                mk = mm_g[sl, MO[1]:MO[1] + 108]
                mask = torch.cat((mk[:n1, :54], mk[n1:, 54:]))
            run(1, sl, mask, 1, two_hot(n1, nr))
        for i, (lo, w, col) in {2: (MO[2], 73, 2), 3: (MO[3], 19, 3), 4: (MO[4], 5, 4), 11: (MO[11], 5, 17)}.items():
            if i in at:
                run(i, at[i], [(at[i].stop - at[i].start, lo, None)] if bits else mm_g[at[i], lo:lo + w], col)
        if 5 in at:
            run(5, at[5], [(at[5].stop - at[5].start, MO[5], None)] if bits else mm_g[at[5], MO[5]:MO[5] + 2], 5, custom=trade[sets[5]].to(main.dtype))
        if 6 in at:        # relative player (propose: mask row 0, steal: row 1)
            sl, n1 = at[6], rp.numel()
            nr = sl.stop - sl.start
            if bits:
                mask = [(n1, MO[6], None), (nr - n1, MO[6] + 3, None)]
            else:
                mk = mm_g[sl, MO[6]:MO[6] + 6]
                mask = torch.cat((mk[:n1, :3], mk[n1:, 3:]))
            run(6, sl, mask, 6, two_hot(n1, nr))
        if 7 in at:        # the recurrent give / receive lists of a proposed trade
            sl, cr = at[7], cur_res[rp]
            give_out, _, lp7, e7 = self._recurrent(H[7], pre[7], None, cr, True, ag[sl, 7:11], False, None)
            filt7 = (lp7 == 0).float()                                           # action_heads_module.py:175
            _, _, lp8, e8 = self._recurrent(H[8], pre[8], give_out * (1 - filt7)[:, None], cr, False, ag[sl, 11:15], False, None)
            lps.append((sl, lp7 + lp8)); ents.append(e7 + e8)
        n_ex, n_yop = rex.numel(), ryop.numel()
        for i in (9, 10):  # resources of an exchange / a Year of Plenty or Monopoly card
            if i not in at:
                continue
            sl = at[i]
            n = sets[i].numel()
            x = torch.zeros(n, 4, device=dev)
            x[:n_ex, 1] = 1.0; x[n_ex:, 0] = 1.0                                  # (play dev, exchange)
            x[n_ex:n_ex + n_yop, 2] = 1.0                                         # (card is YoP, card is Monopoly)
            if i == 9:
                x[n_ex + n_yop:, 3] = 1.0
                # exchange: row 0; a card: row 1 (all ones in the env's masks) x the card's row (Monopoly 2, YoP 3)
                if bits:
                    mask = [(n_ex, MO[9], None), (n_yop, MO[9] + 5, MO[9] + 15), (n - n_ex - n_yop, MO[9] + 5, MO[9] + 10)]
                else:
                    m9 = mm_g[sl, MO[9]:MO[9] + 20].reshape(-1, 4, 5)
                    mask = torch.cat((m9[:n_ex, 0], m9[n_ex:n_ex + n_yop, 1] * m9[n_ex:n_ex + n_yop, 3], m9[n_ex + n_yop:, 1] * m9[n_ex + n_yop:, 2]))
                run(9, sl, mask, 15, x)
            else:
                run(10, sl, [(n, MO[10], None)] if bits else mm_g[sl, MO[10]:MO[10] + 5], 16, torch.cat((x, F.one_hot(ag[sl, 15], 5).float()), -1))
        # every segment of all_rows got exactly one log-prob vector: ONE concatenation in segment order (not a zero fill + a slice copy per
        # head), one index_add back to the batch rows; the entropies as one sum over one concatenation (not a reduction per head)
        lps.sort(key=lambda t: t[0].start)
        assert sum(lp.numel() for _, lp in lps) == all_rows.numel()
        logp = logp0.index_add(0, all_rows, torch.cat([lp for _, lp in lps]))
        return actions.clone(), logp, torch.cat([e0] + ents).sum() / B

    def _recurrent(self, head, pre, fixed, cur_res, from_hand, acts, deterministic, generator):
        """RecurrentResourceActionHead.forward (action_heads_module.py:258-329) without the final type mask.
        pre: the head's trunk contribution (constant over the four steps); fixed: conditioning columns before `out` or None."""
        if acts is not None and RECURRENT_BATCHED:
            return self._recurrent_given(head, pre, fixed, cur_res, from_hand, acts)
        B, x = pre.shape[0], pre
        out = torch.zeros(B, 6, device=x.device, dtype=torch.float32)
        res = cur_res.clone()
        mask = (res > 0).float() if from_hand else torch.ones_like(res)
        mask[:, 0] = (res.sum(-1) == 0).float()
        logp_sum = torch.zeros(B, device=x.device)
        ent_sum = torch.zeros(B, device=x.device)
        chosen = []
        fused = acts is None and getattr(self, "_fused_now", False) and nn_kernels.head_fused_supported(pre)
        for i in range(4):
            cond = out if fixed is None else torch.cat((fixed, out), -1)
            if fused:
                a, step_lp = nn_kernels.head_sample(head, self.D, pre, cond, mask, deterministic, generator)
                ent = None
            else:
                a, step_lp, ent = _categorical(head.logits(pre, cond), mask,
                                               None if acts is None else acts[:, i], deterministic, generator)
            onehot = F.one_hot(a, 6).float()
            out = out + onehot
            res = torch.clamp(res - onehot, min=0)
            mask = (res > 0).float() if from_hand else torch.ones_like(res)
            mask[:, 0] = 1.0
            if i > 0:
                keep = (chosen[-1] > 0).float()
                step_lp = step_lp * keep
                if acts is not None:
                    ent = ent * keep
            logp_sum = logp_sum + step_lp
            if acts is not None:                        # (sampling discards the entropy)
                ent_sum = ent_sum + ent
            chosen.append(a)
            out = torch.cat((torch.zeros_like(out[:, :1]), out[:, 1:]), 1)     # column 0 ("stop") never feeds back; no host constant
        return out, torch.stack(chosen, 1), logp_sum, ent_sum

    def _recurrent_given(self, head, pre, fixed, cur_res, from_hand, acts):
        """`_recurrent` for GIVEN actions (the PPO update) in ONE pass over 4 B rows instead of four passes over B: with the picks known,
        every step's conditioning columns (`out`: how often each resource was picked before, column 0 cleared), hand (`res`: the start
        hand minus the earlier picks, clamped at 0 step by step = clamped once) and mask are functions of acts[:, :i] alone - the
        recurrence of action_heads_module.py:258-329 only exists while sampling.  Same values as the loop (rows are independent; the
        step's log-prob / entropy count only behind a non-stop pick, :306-312); a quarter of its launches, forward and backward."""
        B = pre.shape[0]
        if pre.is_cuda and acts.dtype == torch.int64 and pre.dtype in (torch.bfloat16, torch.float32):
            # every input of the one evaluation from ONE kernel (catan_recurrent_given) instead of ~45 small launches (one_hot, cumsum, clamp,
            # comparisons, concatenations, repeats: 0.3 ms of a minibatch step for the two trade heads)
            cond, mask, given, keep, out_final = nn_kernels.recurrent_given(acts, cur_res, fixed, from_hand, pre.dtype)
            _, lp, ent = _categorical(head.logits(pre.repeat(4, 1), cond), mask, given, False, None)
            return out_final, acts[:, :4], (lp.reshape(4, B).t() * keep).sum(1), (ent.reshape(4, B).t() * keep).sum(1)
        onehot = F.one_hot(acts[:, :4], 6).float()                                   # [B, 4, 6]
        before = torch.cumsum(onehot, 1) - onehot                                     # picks before step i
        res = torch.clamp(cur_res[:, None, :] - before, min=0)                        # [B, 4, 6]
        mask = (res > 0).float() if from_hand else torch.ones_like(res)
        first0 = (cur_res.sum(-1) == 0).float()                                       # step 0: "stop" only with an empty hand
        mask = torch.cat((torch.cat((first0[:, None, None], torch.ones(B, 3, 1, device=pre.device)), 1), mask[:, :, 1:]), 2)
        out = torch.cat((torch.zeros(B, 4, 1, device=pre.device), before[:, :, 1:]), 2)          # column 0 ("stop") never feeds back
        steps = lambda t: t.transpose(0, 1).reshape((4 * B,) + t.shape[2:])           # step-major rows: [step 0 rows, step 1 rows, ...]
        cond = steps(out) if fixed is None else torch.cat((fixed.repeat(4, 1), steps(out)), -1)
        _, lp, ent = _categorical(head.logits(pre.repeat(4, 1), cond), steps(mask), steps(acts[:, :4]), False, None)
        keep = torch.cat((torch.ones(B, 1, device=pre.device), (acts[:, :3] > 0).float()), 1)    # a step counts only behind a non-stop pick
        logp_sum = (lp.reshape(4, B).t() * keep).sum(1)
        ent_sum = (ent.reshape(4, B).t() * keep).sum(1)
        total = before[:, 3] + onehot[:, 3]
        out_final = torch.cat((torch.zeros(B, 1, device=pre.device), total[:, 1:]), 1)
        return out_final, acts[:, :4], logp_sum, ent_sum

    def forward(self, main, masks, cur_res, trade, actions=None, deterministic=False, generator=None, forced_type=None, grouping=None, value_fn=None):
        """main [B,512] (+ lstm_size with the LSTM); masks [B,325]; cur_res [B,6]; trade [B,12]; actions int64 [B,18] or None.
        forced_type int64 [B] or None: rows with a value >= 0 take that action type instead of sampling the type head
        (`condition_on_action_type`, action_heads_module.py:37-48: the type head is skipped, its output is the one-hot).
        -> actions [B,18], joint log-prob [B], entropy (scalar, action_heads_module.py:159-160,174)."""
        if actions is not None and forced_type is None and self.compact_evaluate and main.shape[0] >= self.compact_min_rows:
            return self._evaluate_compact(main, masks, cur_res, trade, actions, grouping, value_fn)
        if isinstance(masks, PackedActionMasks):
            masks = masks.dense()
        self.last_value = value_fn(main) if value_fn is not None else None      # (value_fn: the caller's other consumer of `main`, see _evaluate_compact)
        B, dev = main.shape[0], main.device
        H = self.action_heads
        given = (lambda i: None) if actions is None else (lambda i: actions[:, i])
        m = masks
        cols = {}                                        # action columns, assembled once at the end (not 18 strided column writes)
        one = torch.ones(B, device=dev)

        # mlp_1 of all twelve heads over the trunk: one [B, D] x [D, 12*128] GEMM (each head's weight columns 0..D-1)
        D = self.D
        pre_all = F.linear(main, torch.cat([h.mlp_1.weight[:, :D] for h in H], 0), torch.cat([h.mlp_1.bias for h in H], 0))
        pre = lambda i: pre_all[:, 128 * i:128 * (i + 1)]

        want_ent = actions is not None                  # sampling (`act`) discards the entropy: ~60 tiny launches of a launch-bound pass
        if actions is None and not deterministic and main.is_cuda:
            generator = nn_kernels.UniformPool(generator, B, 18, dev)    # the 18 draws of a pass from one torch.rand

        # inference on the GPU: a head evaluation is ONE kernel (csrc/catan_heads.hip) instead of ~8 small launches ...
        fused = actions is None and nn_kernels.head_fused_supported(pre_all)
        self._fused_now = fused
        if fused and nn_kernels.chained_heads_enabled:
            # ... and the glue between the evaluations (type-conditional mask rows, conditioning columns, log-prob masks, the
            # trade heads' lists: ~200 small torch launches) runs inside those kernels on a per-row state: eighteen launches
            acts, lp = nn_kernels.heads_chain(H, D, pre_all, m, cur_res, trade, deterministic, generator, forced_type)
            return acts, lp, 0.0

        def run(i, extra, mask, idx, count, custom=None):
            if fused:
                cond = extra
                if custom is not None:
                    cf = _ln(H[i].custom_norm, _lin(custom, H[i].custom_mlp.weight, H[i].custom_mlp.bias), relu=True)
                    cond = cf if extra is None else torch.cat((extra, cf.float()), -1)
                a, lpa = nn_kernels.head_sample(H[i], D, pre(i), cond, mask, deterministic, generator)
                return a, lpa * count, 0.0
            a, lpa, ent = _categorical(H[i].logits(pre(i), extra, custom), mask, given(idx), deterministic, generator)
            return a, lpa * count, ((count * ent).mean() if want_ent else 0.0)

        # head 0: action type
        typ, logp, entropy = run(0, None, m[:, MO[0]:MO[0] + 13], 0, one)
        if forced_type is not None:
            forced = forced_type >= 0
            typ = torch.where(forced, forced_type, typ)
            logp = torch.where(forced, torch.zeros_like(logp), logp)
        cols[0] = typ
        _is = {}

        def is_(t):                                      # (typ == t) as float, computed once per type
            if t not in _is:
                _is[t] = (typ == t).float()
            return _is[t]
        # Every (typ == t) the branches share is produced HERE, on the current stream, before the fork: a value first computed
        # (and cached) inside one branch and read by another would cross streams without an event edge between them - in the
        # captured hipGraph the reader would have no dependency on its producer.
        for t in (T_SETTLE, T_CITY, T_ROAD, T_ROBBER, T_RESPOND, T_PROPOSE, T_STEAL, T_DISCARD, T_PLAYDEV, T_EXCHANGE):
            is_(t)
        # Everything below depends on the sampled type only (heads 9 / 10 also on the card of head 4, head 8 on head 7): four
        # independent chains, forked onto side streams in inference (see _Branches); the results are combined after the join.
        br = _HEAD_BRANCHES.fork(main)
        res = {}
        with br.on(1):       # heads 1, 2, 3: corner (conditioned on (settlement, city); mask row by type, build_agent_model.py:113-115), edge, tile
            row = torch.where(typ == T_SETTLE, 0, torch.where(typ == T_CITY, 1, 2))
            cm = m[:, MO[1]:MO[1] + 162].reshape(B, 3, 54).gather(1, row[:, None, None].expand(B, 1, 54)).squeeze(1)
            x = torch.stack((is_(T_SETTLE), is_(T_CITY)), -1)
            res[1] = br.keep(*run(1, x, cm, 1, is_(T_SETTLE) + is_(T_CITY)))
            res[2] = br.keep(*run(2, None, m[:, MO[2]:MO[2] + 73], 2, is_(T_ROAD)))
            res[3] = br.keep(*run(3, None, m[:, MO[3]:MO[3] + 19], 3, is_(T_ROBBER)))
        with br.on(2):       # heads 5, 6, 11: trade response, relative player (conditioned on (propose, steal)), discard
            res[5] = br.keep(*run(5, None, m[:, MO[5]:MO[5] + 2], 5, is_(T_RESPOND), custom=trade.to(main.dtype)))
            row = torch.where(typ == T_PROPOSE, 0, torch.where(typ == T_STEAL, 1, 2))
            pm = m[:, MO[6]:MO[6] + 9].reshape(B, 3, 3).gather(1, row[:, None, None].expand(B, 1, 3)).squeeze(1)
            x = torch.stack((is_(T_PROPOSE), is_(T_STEAL)), -1)
            res[6] = br.keep(*run(6, x, pm, 6, is_(T_PROPOSE) + is_(T_STEAL)))
            res[11] = br.keep(*run(11, None, m[:, MO[11]:MO[11] + 5], 17, is_(T_DISCARD)))
        with br.on(3):       # heads 4 -> 9 -> 10: development card, then resource A / B conditioned on (play dev, exchange) and on the card (YoP, Monopoly)
            card, lp4, e4 = run(4, None, m[:, MO[4]:MO[4] + 5], 4, is_(T_PLAYDEV))
            res[4] = br.keep(card, lp4, e4)
            playdev = typ == T_PLAYDEV
            tcond = torch.stack((is_(T_PLAYDEV), is_(T_EXCHANGE)), -1)
            ccond = torch.stack(((card == C_YOP).float(), (card == C_MONO).float()), -1) * playdev.float()[:, None]   # filtered when head 4 is masked out
            m9 = m[:, MO[9]:MO[9] + 20].reshape(B, 4, 5)
            row_t = torch.where(typ == T_EXCHANGE, 0, 1)
            row_c = torch.where(card == C_MONO, 2, torch.where(card == C_YOP, 3, 1))
            mask_t = m9.gather(1, row_t[:, None, None].expand(B, 1, 5)).squeeze(1)
            mask_c = m9.gather(1, row_c[:, None, None].expand(B, 1, 5)).squeeze(1)
            mask9 = mask_t * torch.where(playdev[:, None], mask_c, torch.ones_like(mask_c))
            cnt9 = (is_(T_PLAYDEV) + is_(T_EXCHANGE)) * torch.where(playdev, ((card == C_YOP) | (card == C_MONO)).float(), one)
            x = torch.cat((tcond, ccond), -1)
            ra, lp9, e9 = run(9, x, mask9, 15, cnt9)
            res[9] = br.keep(ra, lp9, e9)
            cnt10 = (is_(T_PLAYDEV) + is_(T_EXCHANGE)) * torch.where(playdev, (card == C_YOP).float(), one)
            x = torch.cat((x, F.one_hot(ra, 5).float() * (cnt9 != 0).float()[:, None]), -1)
            res[10] = br.keep(*run(10, x, m[:, MO[10]:MO[10] + 5], 16, cnt10))
        with br.on(0):       # heads 7 -> 8: the recurrent give / receive resource lists (the longest chain: eight sequential draws)
            prop = is_(T_PROPOSE)
            give_out, give_a, lp7, e7 = self._recurrent(H[7], pre(7), None, cur_res, True, None if actions is None else actions[:, 7:11], deterministic, generator)
            lp7 = lp7 * prop
            filt7 = (lp7 == 0).float()                                           # action_heads_module.py:175
            _, recv_a, lp8, e8 = self._recurrent(H[8], pre(8), give_out * (1 - filt7)[:, None], cur_res, False,
                                                 None if actions is None else actions[:, 11:15], deterministic, generator)
        br.join()
        lps = [logp, lp7, lp8 * prop]
        for i, col in ((1, 1), (2, 2), (3, 3), (4, 4), (5, 5), (6, 6), (9, 15), (10, 16), (11, 17)):
            a, lp, e = res[i]
            cols[col] = a; lps.append(lp); entropy = entropy + e
        logp = torch.stack(lps).sum(0)                   # (one reduction instead of eleven adds)
        cols[7] = give_a; entropy = entropy + ((e7 * prop).mean() if want_ent else 0.0)
        cols[11] = recv_a; entropy = entropy + ((e8 * prop).mean() if want_ent else 0.0)
        out = torch.cat((torch.stack([cols[i] for i in range(7)], 1), cols[7], cols[11], torch.stack([cols[15], cols[16], cols[17]], 1)), 1)
        return out, logp, entropy


class CatanPolicy(nn.Module):
    """SettlersAgentPolicy (RL/models/policy.py:12-111) for flat batched inputs.

    include_lstm (off in the reference's defaults, build_agent_model.py:26): a one-layer LSTM(512 -> lstm_size) over each
    seat's successive decisions; its output is concatenated to the 512-wide trunk for the value MLP and every action head
    (policy.py:36-45,59-66).  `hidden` = (h, c), each [rows, lstm_size]; `nonterminal` [rows] or [rows,1] multiplies the
    incoming state (policy.py:113-166)."""

    VALUE_MEAN, VALUE_STD = 150.0, 150.0          # ValueFunctionNormaliser(mean=150, std=150), policy.py:23
    use_value_normalisation = True

    def __init__(self, include_lstm=False, lstm_size=256):
        super().__init__()
        self.include_lstm, self.lstm_size = bool(include_lstm), int(lstm_size)
        self.observation_module = _ObservationModule()
        D = 512
        if self.include_lstm:
            self.lstm = nn.LSTM(num_layers=1, hidden_size=self.lstm_size, input_size=D, batch_first=False)   # policy.py:38-43
            for name, prm in self.lstm.named_parameters():
                if "bias" in name:
                    nn.init.zeros_(prm)
                else:
                    nn.init.orthogonal_(prm)
            D += self.lstm_size
        self.action_head_module = _ActionHeads(D)
        self.value_network_fc_1 = nn.Linear(D, 256)
        self.value_network_fc_2 = nn.Linear(256, 128)
        self.value_out = nn.Linear(128, 1)
        self.v_norm_1 = nn.LayerNorm(256)
        self.v_norm_2 = nn.LayerNorm(128)

    # ---- pieces
    def initial_hidden(self, rows, device=None):
        """The zero state a seat starts a game with (game_manager.py:54-59,121-124)."""
        dev = self.value_out.weight.device if device is None else device
        return (torch.zeros(rows, self.lstm_size, device=dev), torch.zeros(rows, self.lstm_size, device=dev))

    def _forward_lstm(self, x, hidden, nonterminal):
        """`_forward_lstm` (policy.py:113-166).  x [R,512]; hidden (h, c) with R rows -> one step per row; with B < R rows
        -> x is T = R/B steps of B sequences, time-major (row t*B + b), and the state entering step t is multiplied by
        nonterminal[t*B + b].  (The reference runs the stretches between zero masks through one nn.LSTM call each and
        multiplies by the mask at their first step only: the same recurrence, since the other masks are 1.)
        The input projection of all steps is one GEMM; the recurrence is h @ W_hh^T plus the gate arithmetic per step."""
        h, c = hidden
        L = self.lstm_size
        R, B = x.shape[0], h.shape[0]
        if R % B != 0:
            raise ValueError(f"LSTM input rows ({R}) are not a multiple of the hidden-state rows ({B})")
        T = R // B
        m = nonterminal.reshape(T, B, 1).to(torch.float32)
        w_ih, w_hh = self.lstm.weight_ih_l0, self.lstm.weight_hh_l0
        gx = F.linear(x, w_ih, self.lstm.bias_ih_l0 + self.lstm.bias_hh_l0).reshape(T, B, 4 * L)
        h, c = h.float(), c.float()
        outs = []
        fused = nn_kernels.lstm_cell_supported(gx, c)            # GPU: the gate arithmetic of a step is one HIP kernel
        gxs = gx.unbind(0)                                       # (one stack in the backward instead of T zero-padded slices)
        for t in range(T):
            h = h * m[t]
            gh = F.linear(h.to(x.dtype), w_hh)
            if fused:
                h, c = nn_kernels.lstm_cell(gxs[t], gh.to(gx.dtype), c, m[t])
            else:
                g = gxs[t].float() + gh.float()
                i, f, gg, o = g[:, :L], g[:, L:2 * L], g[:, 2 * L:3 * L], g[:, 3 * L:]    # torch.nn.LSTM gate order i, f, g, o
                c = torch.sigmoid(f) * (c * m[t]) + torch.sigmoid(i) * torch.tanh(gg)
                h = torch.sigmoid(o) * torch.tanh(c)
            outs.append(h)
        out = outs[0] if T == 1 else torch.stack(outs, 0).reshape(R, L)
        return out.to(x.dtype), (h, c)

    def base(self, obs_f, lists, lens, hidden=None, nonterminal=None, tile_features=None, tile_dedupe=None):
        """-> (value [B,1] fp32, main [B,512(+lstm_size)], hidden or None)   (policy.py:59-66)"""
        main, hidden = self._main(obs_f, lists, lens, hidden, nonterminal, tile_features, tile_dedupe)
        return self._value(main), main, hidden

    def _main(self, obs_f, lists, lens, hidden=None, nonterminal=None, tile_features=None, tile_dedupe=None):
        main = self.observation_module(obs_f, lists, lens, tile_features, tile_dedupe)
        if self.include_lstm:
            if hidden is None:
                raise ValueError("include_lstm: hidden=(h, c) and nonterminal are required")
            if nonterminal is None:
                nonterminal = torch.ones(main.shape[0], device=main.device)
            out, hidden = self._forward_lstm(main, hidden, nonterminal)
            main = torch.cat((main, out), -1)
        return main, hidden

    def _value(self, main):
        v = _lin(_ln(self.v_norm_2, _lin(_ln(self.v_norm_1, _lin(main, self.value_network_fc_1.weight, self.value_network_fc_1.bias), relu=True), self.value_network_fc_2.weight, self.value_network_fc_2.bias), relu=True),
                 self.value_out.weight, self.value_out.bias)
        return v.float()

    @staticmethod
    def _custom(obs_f):
        if isinstance(obs_f, ObsParts):
            obs_f = obs_f.head
        return obs_f[:, 12:18].float(), obs_f[:, 0:12].float()      # current_resources, proposed_trade

    # ---- reference-shaped API (with include_lstm the new hidden state is returned as a last extra item)
    def act(self, obs_f, lists, lens, masks, deterministic=False, generator=None, condition_on_action_type=None,
            hidden=None, nonterminal=None):
        """condition_on_action_type: int64 [B] (entries < 0 = free) or None (RL/models/policy.py:72-82)."""
        main, hidden = self._main(obs_f, lists, lens, hidden, nonterminal)
        br = _VALUE_BRANCHES.fork(main)          # inference: the value head (three products, two LayerNorms: ~120 us in a row at 65 536 rows) beside the action heads
        with br.on(1):
            value = br.keep(self._value(main))
        cur_res, trade = self._custom(obs_f)
        actions, logp, _ = self.action_head_module(main, masks.float(), cur_res, trade, None, deterministic, generator,
                                                   forced_type=condition_on_action_type)
        br.join()
        return (value, actions, logp[:, None], hidden) if self.include_lstm else (value, actions, logp[:, None])

    def evaluate_actions(self, obs_f, lists, lens, masks, actions, hidden=None, nonterminal=None, tile_dedupe=None, grouping=None):
        """grouping: this batch's entry of _ActionHeads.precompute_groupings (the learner computes them for a whole epoch at once)"""
        ahm = self.action_head_module
        if grouping is None or not ahm.wants_grouping(obs_f.shape[0], actions):
            grouping = ahm.start_grouping(actions) if ahm.wants_grouping(obs_f.shape[0], actions) else None    # (before the long forward)
        main, hidden = self._main(obs_f, lists, lens, hidden, nonterminal, tile_dedupe=tile_dedupe)
        cur_res, trade = self._custom(obs_f)
        _, logp, entropy = ahm(main, masks.float(), cur_res, trade, actions, grouping=grouping, value_fn=self._value)
        value, ahm.last_value = ahm.last_value, None
        return (value, logp[:, None], entropy, hidden) if self.include_lstm else (value, logp[:, None], entropy)

    def get_value(self, obs_f, lists, lens, hidden=None, nonterminal=None, tile_features=None):
        return self.base(obs_f, lists, lens, hidden, nonterminal, tile_features)[0]

    def inference_copy(self, dtype=torch.bfloat16):
        """A no-grad copy for acting under `torch.autocast(dtype)`: the Linear / LSTM / embedding parameters are stored in
        `dtype` already, so autocast has nothing to cast (with fp32 masters it re-casts every weight and bias on every call:
        ~200 tiny launches of the ~800 of an `act`, which is launch-bound).  LayerNorm parameters stay fp32 (the HIP
        LayerNorm kernels read them as fp32).  Refresh with `load_from(master)` after an optimiser step."""
        import copy
        c = copy.deepcopy(self).eval().requires_grad_(False)
        c._inference_dtype = dtype
        # (the dev-card list modules - the 6-row embedding and the two 16-wide attentions - are evaluated in fp32 from
        # small tables whatever the autocast dtype, _card_summary: their parameters stay fp32)
        fp32_prefixes = ("observation_module.dev_card_embedding", "observation_module.hidden_card_mha", "observation_module.played_card_mha")
        for name, mod in c.named_modules():
            if isinstance(mod, (nn.Linear, nn.Embedding)) and not name.startswith(fp32_prefixes):
                mod.to(dtype)
            elif isinstance(mod, nn.LSTM):         # its two bias vectors are added in fp32 before the cast (_forward_lstm)
                for name, prm in mod.named_parameters():
                    if name.startswith("weight"):
                        prm.data = prm.data.to(dtype)
        return c

    def refresh_kernel_packs(self):
        """Brings the parameter packs the fused kernels read (the tile encoder's, nn_kernels.tile_encoder_pack) up to date, in
        place - what a forward would do on its way; a captured hipGraph replay (GraphedAct) does not run that host code."""
        te = self.observation_module.tile_encoder
        if next(te.parameters()).is_cuda and getattr(te, "_fused_pack", None) is not None:
            nn_kernels.tile_encoder_pack(te)
        nn_kernels.refresh_card_tables(self.observation_module)
        for h in self.action_head_module.action_heads:          # the fused head kernels' packs (nn_kernels.head_pack)
            cache = getattr(h, "_fused_pack", None)
            if cache is not None and h.mlp_2.weight.is_cuda:
                nn_kernels.head_pack(h, cache[0][3])
            if getattr(h, "_custom_pack", None) is not None and h.mlp_2.weight.is_cuda:
                nn_kernels.head5_custom_pack(h)

    @torch.no_grad()
    def load_from(self, master):
        """Copies (and casts) the parameters of `master` into this copy."""
        for (k, dst), (k2, src) in zip(self.state_dict().items(), master.state_dict().items()):
            assert k == k2
            dst.copy_(src)

    def denormalise(self, v):
        return self.VALUE_MEAN + v * self.VALUE_STD          # RL/models/utils.py:20-21

    # entries of the reference `SettlersAgentPolicy.state_dict()` that hold no learnable state: the empty `dummy_param`
    # device probes of its sub-modules and the value normaliser's constants (RL/models/utils.py:9-15, policy.py:23)
    REFERENCE_DUMMY_KEYS = (
        "dummy_param", "value_normaliser.dummy_param", "observation_module.hidden_card_mha.dummy_param",
        "observation_module.played_card_mha.dummy_param",
        "observation_module.tile_encoder.encoder_layers.0.multi_headed_attention.dummy_param",
        "observation_module.tile_encoder.encoder_layers.1.multi_headed_attention.dummy_param",
        "observation_module.current_player_module.dummy_param", "observation_module.other_players_module.dummy_param",
        "action_head_module.dummy_param", "action_head_module.action_heads.7.dummy_param",
        "action_head_module.action_heads.8.dummy_param")

    @classmethod
    def to_reference_state_dict(cls, sd):
        """A `CatanPolicy.state_dict()` completed to the key set of the reference net, so that the reference's strict
        `central_policy.load_state_dict` / `rollout_manager.update_policy` (robust_train.py:55-59, game_manager.py:161-162)
        accept it."""
        out = {k: v.detach().cpu() for k, v in sd.items()}
        for k in cls.REFERENCE_DUMMY_KEYS:
            out.setdefault(k, torch.empty(0))
        out.setdefault("value_normaliser.mean", torch.tensor([cls.VALUE_MEAN], dtype=torch.float32))
        out.setdefault("value_normaliser.std", torch.tensor([cls.VALUE_STD], dtype=torch.float32))
        return out

    def reference_state_dict(self):
        return self.to_reference_state_dict(self.state_dict())

    def load_reference_state_dict(self, sd):
        """Loads a reference `SettlersAgentPolicy.state_dict()` (same parameter names; the reference's empty
        `dummy_param` entries and the value normaliser constants are ignored)."""
        own = self.state_dict()
        filt = {k: v for k, v in sd.items() if k in own}
        missing = [k for k in own if k not in filt]
        if missing:
            raise KeyError(f"reference state_dict lacks {missing[:5]} ...")
        self.load_state_dict(filt, strict=True)
