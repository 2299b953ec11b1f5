"""`roofline_learner` of bench.py: the learner-side kernels of a PPO update (BASELINE config 3), each timed ALONE with HIP
events on the stream it is launched on, at the shapes of the real update (65 536 games, T = 200, 204 800-row minibatches), against
the HBM roofline on its ALGORITHMIC bytes (the tensors it must read and write once, stated per kernel below) and - where it runs
on the matrix cores - its MFMA FLOP/s against the dense bf16 peak.

Kernels: the two fused learner kernels north_star names (k_gae, k_ppo_loss), the observation encoder of a rollout pass
(k_obs_rows), the fused tile encoder of every inference pass and of the training forward (k_tile_encoder_fwd), the one-pass
backward kernels of its sub-layers (k_ffn_bwd_dx, k_qkv_bwd_dx), the row gathers of a minibatch, one fused action-head evaluation
(k_head_fwd), and the longest other hand-written kernels of a minibatch step by the kernel trace (profiles/r03_train_step_kernel_stats.csv):
the tile encoder's attention backward, a row product, a LayerNorm backward and a weight gradient.  The library GEMMs of the
step are not listed: they are rocBLAS / hipBLASLt code."""
import torch

HBM_PEAK_GBS = 8000.0
MFMA_PEAK_TFLOPS = 2500.0          # dense bf16 (MI355X_MICROARCH.md)


def _time_us(fn, reps=10, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


def _entry(kernel, what, us, nbytes, flops=None, note=None):
    gbs = nbytes / (us * 1e-6) / 1e9
    e = {"kernel": kernel, "shape": what, "avg_us": us, "algorithmic_bytes": nbytes, "achieved_gbs": gbs, "hbm_frac": gbs / HBM_PEAK_GBS, "bound": "hbm"}
    if flops is not None:
        tf = flops / (us * 1e-6) / 1e12
        e.update(mfma_flops=flops, achieved_tflops=tf, mfma_frac=tf / MFMA_PEAK_TFLOPS)
    if note:
        e["note"] = note
    return e


def learner_rooflines(env, net, T=200, rows_mb=204800):
    """env: VecCatanEnv with n games (mid-game states); net: CatanPolicy on the device (fp32 master).  -> list of dicts."""
    from settlers_of_catan_rl_amd import nn_kernels, ppo as K, spec
    out = []
    dev, n = env.device, env.n
    g = torch.Generator(device=dev).manual_seed(0)
    # ---- k_gae + k_adv_stats + k_adv_normalise: 3 reads + 2 writes of (T, n) fp32 = 20 B per (t, game); normalise: read + write = 8 B
    r = torch.randn(T, n, device=dev, generator=g); v = torch.randn(T + 1, n, device=dev, generator=g); m = (torch.rand(T + 1, n, device=dev, generator=g) > 0.02).float()
    out.append(_entry("k_gae + k_adv_stats + k_adv_normalise", f"T = {T} x {n} games", _time_us(lambda: K.compute_gae(r, v, m, 0.999, 0.95, process_group=False)),
                      28 * T * n, note="rewards, values, masks in; returns, advantages out (20 B per (t, game)); the normalisation reads and writes the advantages once more (8 B)"))
    del r, v, m
    # ---- k_ppo_loss: 6 fp32 streams in, 2 gradient streams out = 32 B per row
    x = [torch.randn(rows_mb, device=dev, generator=g) for _ in range(6)]
    out.append(_entry("k_ppo_loss (forward + analytic gradients)", f"{rows_mb} rows", _time_us(lambda: K.ppo_loss(x[0] - 3, x[1], x[2] - 3, x[3], x[4], x[5], 0.2, 1.0, value_normaliser=(150.0, 150.0))),
                      32 * rows_mb, note="launch-sized problem (6.5 MB): the time is the launch + the 256-workgroup reduction, not bandwidth"))
    del x
    # ---- k_obs_rows: record in (704 B), bf16 observation row (3 574 B) + int32 card lists / lengths (520 B) out per game;
    #      + the rollout-storage rows of the games whose active seat decides (one in four: 3 574 + 130 B each)
    dense = env.get_obs_rows(torch.bfloat16)
    st_f = torch.zeros((4, n, spec.OBS_FLOATS), dtype=torch.bfloat16, device=dev); st_l = torch.zeros((4, n, 5, 25), dtype=torch.int8, device=dev)
    st_n = torch.zeros((4, n, 5), dtype=torch.int8, device=dev)
    t = torch.randint(0, 4, (n,), device=dev, generator=g); sel = torch.rand(n, device=dev, generator=g) < 0.25
    out.append(_entry("k_obs_rows (bf16 dense)", f"{n} games", _time_us(lambda: env.get_obs_rows(torch.bfloat16, out=dense), reps=30), (704 + 3574 + 520) * n))
    frac = float(sel.float().mean())
    out.append(_entry("k_obs_rows (bf16 dense + rollout-storage rows of 1 game in 4)", f"{n} games",
                      _time_us(lambda: env.get_obs_rows(torch.bfloat16, out=dense, rows=(st_f, st_l, st_n), t=t, sel=sel), reps=30),
                      int((704 + 3574 + 520 + 9 + frac * (3574 + 130)) * n)))
    del st_f, st_l, st_n
    # ---- k_tile_encoder_fwd: 19 x 60 bf16 in (2 280 B), 19 x 25 bf16 out (950 B) per board; MFMA work per board: 19 tokens x
    #      (64x64 + 2 x (64x192 + 64x64 + 64x128 + 128x64) + 64x32) MACs + 2 layers x 4 heads x 2 products of 32x32x16 (padded)
    te = net.observation_module.tile_encoder
    boards = rows_mb
    tiles = (torch.rand(boards, 19, 60, device=dev, generator=g) < 0.1).to(torch.bfloat16)
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
        us = _time_us(lambda: nn_kernels.tile_encoder_forward(te, tiles), reps=5)
    macs = 19 * (64 * 64 + 2 * (64 * 192 + 64 * 64 + 64 * 128 + 128 * 64) + 64 * 32) + 2 * 4 * 2 * 32 * 32 * 16
    out.append(_entry("k_tile_encoder_fwd (whole tile encoder, inference)", f"{boards} boards", us, (2280 + 950) * boards, flops=2 * macs * boards,
                      note="VALU / L2-latency bound (LayerNorm, softmax, epilogues), not HBM- or MFMA-bound: DESIGN.md 4.5 (iv)"))
    # ---- the same kernel as the TRAINING forward: it also stores what the backward kernels read (1 273 bf16 per token: 48 KB per board;
    #      the LayerNorm outputs n1 / n2 are recomputed by the backward passes instead)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        us = _time_us(lambda: nn_kernels.tile_encoder_train(te, tiles), reps=5)
    out.append(_entry("k_tile_encoder_fwd<SAVE> (training forward: + the activations the backward reads)", f"{boards} boards", us,
                      (2280 + 950 + 19 * 1273 * 2) * boards, flops=2 * macs * boards, note="includes the per-step re-pack of the encoder's weights (~60 small launches)"))
    del tiles
    # ---- the backward of the encoder's sub-layers up to the weight gradients, one pass each (csrc/catan_te_bwd.hip)
    tokf = boards * 19
    dxg = torch.randn(tokf, 64, device=dev, generator=g).to(torch.bfloat16); hh = torch.relu(torch.randn(tokf, 128, device=dev, generator=g)).to(torch.bfloat16)
    xm = torch.randn(tokf, 64, device=dev, generator=g).to(torch.bfloat16)
    w2t = torch.randn(128, 64, device=dev, generator=g).to(torch.bfloat16); w1t = torch.randn(64, 128, device=dev, generator=g).to(torch.bfloat16)
    lw = torch.ones(64, device=dev); dh = torch.empty_like(hh); dxo = torch.empty_like(xm); dl = torch.zeros(2, 64, device=dev)
    from settlers_of_catan_rl_amd import _lib
    P, S = nn_kernels._ptr, nn_kernels._stream
    us = _time_us(lambda: _lib.check(_lib.lib().catan_ffn_bwd_dx(P(dxg), P(hh), P(xm), P(w2t), P(w1t), P(lw), 1e-5, P(dh), P(dxo), P(dl[0]), P(dl[1]), tokf, S())), reps=5)
    out.append(_entry("k_ffn_bwd_dx (masked dX W2, dH W1, LayerNorm backward + residual)", f"{tokf} rows", us, (64 + 128 + 64 + 128 + 64) * 2 * tokf,
                      flops=2 * 2 * 64 * 128 * tokf, note="dX, H, X in; dH, dX' out"))
    dq = torch.randn(tokf, 192, device=dev, generator=g).to(torch.bfloat16); wqt = torch.randn(64, 192, device=dev, generator=g).to(torch.bfloat16)
    us = _time_us(lambda: _lib.check(_lib.lib().catan_qkv_bwd_dx(P(dq), P(xm), P(dxg), P(wqt), P(lw), 1e-5, P(dxo), P(dl[0]), P(dl[1]), tokf, S())), reps=5)
    out.append(_entry("k_qkv_bwd_dx (dQKV Wqkv, LayerNorm backward + residual)", f"{tokf} rows", us, (192 + 64 + 64 + 64) * 2 * tokf, flops=2 * 192 * 64 * tokf))
    # ---- the same chains WITH the sub-layers' weight gradients in the pass (k_ffn_bwd_w, k_qkv_bwd_w: what the update's backward runs)
    oo = torch.randn(tokf, 64, device=dev, generator=g).to(torch.bfloat16); n2 = torch.randn(tokf, 64, device=dev, generator=g).to(torch.bfloat16)
    wot = torch.randn(64, 64, device=dev, generator=g).to(torch.bfloat16)
    do, lb = torch.empty_like(oo), torch.randn(64, device=dev, generator=g)
    accw = torch.zeros(64 * 128 + 64 + 128 * 64 + 128 + 192 * 64 + 192 + 64 * 64 + 64, device=dev)
    # (the update's configuration: n = NULL - the LayerNorm outputs are recomputed from X in the passes, the forward does not store them)
    us = _time_us(lambda: _lib.check(_lib.lib().catan_ffn_outproj_bwd(P(dxg), P(hh), P(xm), None, P(w2t), P(w1t), P(lw), P(lb), 1e-5, P(dxo), P(accw[:8192]),
                                                                      P(accw[8192:8256]), P(accw[8256:16448]), P(accw[16448:16576]), P(dl[0]), P(dl[1]),
                                                                      P(oo), P(wot), P(do), P(accw[29056:33152]), P(accw[33152:33216]), tokf, S())), reps=5)
    out.append(_entry("k_ffn_bwd_w<out-projection> (k_ffn_bwd_dx + dW2, dW1 + the out-projection's dO, dWo in the same pass)", f"{tokf} rows", us,
                      (64 + 128 + 64 + 64 + 64 + 64) * 2 * tokf, flops=(2 * 4 * 64 * 128 + 2 * 2 * 64 * 64) * tokf,
                      note="dX, H, X, O in; dX', dO out; dH and N stay in LDS"))
    us = _time_us(lambda: _lib.check(_lib.lib().catan_qkv_bwd(P(dq), P(xm), P(dxg), None, P(wqt), P(lw), P(lb), 1e-5, P(dxo), P(accw[16576:28864]), P(accw[28864:29056]),
                                                              P(dl[0]), P(dl[1]), tokf, S())), reps=5)
    out.append(_entry("k_qkv_bwd_w<N recomputed> (k_qkv_bwd_dx + dWqkv = dQKV^T N in the same pass)", f"{tokf} rows", us, (192 + 64 + 64 + 64) * 2 * tokf,
                      flops=2 * 2 * 192 * 64 * tokf, note="dQKV, X, the residual gradient in; dX out; N from X"))
    del dxg, hh, xm, dh, dxo, dq, oo, do, n2
    # ---- row movement of a minibatch: the distinct boards' tile features out of the rollout rows (2-byte aligned 3 574-byte rows),
    #      a per-board result spread to the rows, the rows' gradients summed per board
    rows_all = 16 * rows_mb
    store = torch.randn(rows_all // 16, 16 * 1787, device=dev, generator=g).to(torch.bfloat16).view(rows_all, 1787)
    idx = torch.randint(0, rows_all, (boards,), device=dev, generator=g)
    out.append(_entry("k_gather_rows (tile features of the distinct boards)", f"{boards} rows of 2 280 B", _time_us(lambda: nn_kernels.gather_rows(store[:, 18:1158], idx), reps=5),
                      2 * 2280 * boards))
    del store
    U = int(0.875 * rows_mb)
    inv = torch.randint(0, U, (rows_mb,), device=dev, generator=g); inv[:U] = torch.arange(U, device=dev)
    order = torch.argsort(inv, stable=True)
    start = torch.cat((torch.zeros(1, dtype=torch.int64, device=dev), torch.cumsum(torch.bincount(inv, minlength=U), 0)))
    srcu = torch.randn(U, 480, device=dev, generator=g).to(torch.bfloat16).requires_grad_(True)
    y = nn_kernels.expand_rows(srcu, inv, order, start)
    dy = torch.randn_like(y)
    out.append(_entry("k_expand_rows16 (per-board rows of 960 B to the minibatch rows)", f"{U} -> {rows_mb} rows",
                      _time_us(lambda: nn_kernels.expand_rows(srcu.detach(), inv, order, start), reps=5), 960 * (U + rows_mb) + 8 * rows_mb))
    out.append(_entry("k_segment_sum16 (its backward: the rows' gradients summed per board)", f"{rows_mb} -> {U} rows",
                      _time_us(lambda: torch.autograd.grad(y, srcu, dy, retain_graph=True), reps=5), 960 * (U + rows_mb) + 16 * rows_mb))
    del srcu, y, dy
    # ---- k_head_fwd: one action-head evaluation at rollout width: 128 bf16 in (256 B) + mask row (4 K B) + u in, action + logp out (12 B)
    ahm = net.action_head_module
    head = ahm.action_heads[2]                       # the road head: K = 73, no conditioning columns
    pre_all = torch.randn(n, 1536, device=dev, generator=g).to(torch.bfloat16)
    mask = torch.ones(n, 73, device=dev)
    with torch.no_grad():
        us = _time_us(lambda: nn_kernels.head_sample(head, ahm.D, pre_all[:, 256:384], None, mask, deterministic=True), reps=30)
    out.append(_entry("k_head_fwd (LayerNorm + 128x128 + 128x73 + masked categorical)", f"{n} rows, K = 73", us, (256 + 4 * 73 + 12) * n,
                      flops=2 * (128 * 128 + 128 * 80) * n, note="includes the Python wrapper's launch (two allocations); two waves per SIMD, 256 workgroups"))
    del pre_all, mask
    # ---- the hand-written kernels that lead the kernel trace of a minibatch step (tile encoder, 204 800 boards x 19 tokens)
    tok = rows_mb * 19
    qkv = torch.randn(rows_mb, 19, 3, 4, 16, device=dev, generator=g).to(torch.bfloat16).requires_grad_(True)
    o = nn_kernels.small_attention(qkv)
    go = torch.randn_like(o)
    out.append(_entry("k_attn_mfma_fwd (19 x 19 attention, 4 heads x 16)", f"{rows_mb} sequences", _time_us(lambda: nn_kernels.small_attention(qkv.detach()), reps=5),
                      (19 * 192 + 19 * 64) * 2 * rows_mb, flops=2 * 2 * 4 * 19 * 19 * 16 * rows_mb))
    out.append(_entry("k_attn_mfma_bwd", f"{rows_mb} sequences", _time_us(lambda: torch.autograd.grad(o, qkv, go, retain_graph=True), reps=5),
                      (19 * 192 * 2 + 19 * 64) * 2 * rows_mb, flops=2 * 5 * 4 * 19 * 19 * 16 * rows_mb, note="qkv and dout in, dqkv out; the probabilities are recomputed"))
    del qkv, o, go
    x64 = torch.randn(tok, 64, device=dev, generator=g).to(torch.bfloat16)
    w = torch.randn(128, 64, device=dev, generator=g).to(torch.bfloat16); b = torch.zeros(128, device=dev, dtype=torch.bfloat16)
    with torch.no_grad():
        us = _time_us(lambda: nn_kernels.linear_inference(x64, w, b), reps=5)
    out.append(_entry("k_linear_rows (64 -> 128, the FFN's first product)", f"{tok} rows", us, (64 + 128) * 2 * tok, flops=2 * 64 * 128 * tok))
    dy = torch.randn(tok, 128, device=dev, generator=g).to(torch.bfloat16)
    out.append(_entry("k_wgrad_tr (dW = dY^T X, 128 x 64)", f"{tok} rows", _time_us(lambda: nn_kernels.wgrad(x64, dy), reps=5), (64 + 128) * 2 * tok, flops=2 * 64 * 128 * tok))
    del dy, w, b
    # ---- the heads' first layers on their row segments (a minibatch step: eleven segments of the gathered 512-wide trunk rows, 128 outputs
    #      each): one launch per (segment, 128-column slice) = 44 launches, against catan_linear_wgrad_grouped (2 launches)
    seg_rows = (1820, 6101, 4007, 3085, 37138, 40142, 36842, 36842, 13073, 12858, 16298)
    xs = [torch.randn(r_, 512, device=dev, generator=g).to(torch.bfloat16) for r_ in seg_rows]
    dys = [torch.randn(r_, 128, device=dev, generator=g).to(torch.bfloat16) for r_ in seg_rows]
    seg_bytes = sum(r_ * (512 + 128) * 2 for r_ in seg_rows); seg_flops = sum(2 * r_ * 512 * 128 for r_ in seg_rows)
    big = [(x_, d_) for x_, d_ in zip(xs, dys) if x_.shape[0] >= 4096]
    us_single = _time_us(lambda: [nn_kernels.wgrad(x_, d_) for x_, d_ in big], reps=5)
    us_grouped = _time_us(lambda: nn_kernels.wgrad_grouped(big), reps=5)
    gb = sum(x_.shape[0] * (512 + 128) * 2 for x_, _ in big); gf = sum(2 * x_.shape[0] * 512 * 128 for x_, _ in big)
    out.append(_entry("k_wgrad_tr_grouped (the heads' first layers on their row segments, 4 column slices each, in 2 launches)", f"{sum(x_.shape[0] for x_, _ in big)} rows in {len(big)} segments",
                      us_grouped, gb, flops=gf, note=f"the same products as one launch per (segment, slice): {us_single:.0f} us in {4 * len(big)} launches; dY is read once per 128-column slice of X (the algorithmic bytes count it once)"))
    del xs, dys, big
    ln = torch.nn.LayerNorm(64).to(dev)
    xg = x64.clone().requires_grad_(True)
    y = nn_kernels.small_layer_norm(xg, ln, False)
    gy = torch.randn_like(y)
    out.append(_entry("k_lnw_fwd (LayerNorm 64)", f"{tok} rows", _time_us(lambda: nn_kernels.small_layer_norm(x64, ln, False), reps=5), 2 * 64 * 2 * tok))
    out.append(_entry("k_lnw_bwd (LayerNorm 64)", f"{tok} rows", _time_us(lambda: torch.autograd.grad(y, xg, gy, retain_graph=True), reps=5), 3 * 64 * 2 * tok,
                      note="x and dy in, dx out; weight / bias gradients are reduced per block"))
    return out
