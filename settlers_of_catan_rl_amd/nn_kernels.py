"""Hand-written HIP kernels used inside the (otherwise PyTorch) policy net: fused small-sequence attention
(csrc/catan_nn.hip).  torch is plumbing: tensors, streams, autograd registration."""
import ctypes as C

import os
import torch

from . import _lib

SUPPORTED = {(19, 4, 16), (25, 4, 4)}


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _aligned(t):
    """contiguous and 16-byte aligned (the wide LayerNorm uses 16 B global accesses; a contiguous view may start anywhere)"""
    t = t.contiguous()
    return t if t.data_ptr() % 16 == 0 else t.clone()


class _SmallAttention(torch.autograd.Function):
    @staticmethod
    def forward(ctx, qkv, lens):
        B, L, three, H, HD = qkv.shape
        qkv = qkv.contiguous()
        out = torch.empty((B, L, H * HD), dtype=qkv.dtype, device=qkv.device)
        _lib.check(_lib.lib().catan_attention_fwd(_ptr(qkv), _ptr(lens), _ptr(out), B, L, H, HD,
                                                  int(qkv.dtype == torch.bfloat16), _stream()))
        ctx.save_for_backward(qkv, lens)
        return out

    @staticmethod
    def backward(ctx, dout):
        qkv, lens = ctx.saved_tensors
        B, L, three, H, HD = qkv.shape
        dout = dout.contiguous().to(qkv.dtype)
        dqkv = torch.empty_like(qkv)
        _lib.check(_lib.lib().catan_attention_bwd(_ptr(qkv), _ptr(lens), _ptr(dout), _ptr(dqkv), B, L, H, HD,
                                                  int(qkv.dtype == torch.bfloat16), _stream()))
        return dqkv, None


def small_attention(qkv, lens=None):
    """qkv [B, L, 3, H, HD] (float32 or bfloat16, CUDA) -> [B, L, H*HD]; lens int32 [B] masks keys >= len."""
    assert qkv.is_cuda and qkv.dtype in (torch.float32, torch.bfloat16) and tuple(qkv.shape[1:2] + qkv.shape[3:]) in SUPPORTED
    if lens is not None:
        lens = lens.to(torch.int32).contiguous()
    return _SmallAttention.apply(qkv, lens)


def supported(L, H, HD):
    return (L, H, HD) in SUPPORTED


LN_WIDTHS = {16, 25, 32, 64, 128, 256, 512}


class _GradArena(object):
    """The zero-initialised fp32 accumulators of the backward kernels (weight / bias gradients summed over row blocks with
    atomics) as slices of ONE buffer that is cleared by ONE fill per optimiser step: as ~100 `torch.zeros` calls they were ~100
    launches of ~3.5 us each in every minibatch step (profiles/r04_update_step_ops.txt).  `begin_step()` (PPOTrainer, before a
    step's forward) clears what the previous step used and rewinds; outside a step - or past the end of the buffer, which then
    grows for the next step - `zeros` is `torch.zeros`.  The slices become the parameters' `.grad` (autograd takes them over):
    they are read by the optimiser before the next `begin_step`, which is the contract of `GradBucket.zero()` as well."""

    def __init__(self):
        self.buf, self.off, self.used, self.need, self.active = None, 0, 0, 0, False
        self.enabled = True                              # (A/B switch: tools/ab_step_switches.py)

    def begin_step(self, device):
        if not self.enabled:
            self.active = False
            return
        want = max(self.need + self.need // 4, 1 << 20)
        if self.buf is None or self.buf.device != torch.device(device) or self.buf.numel() < want:
            self.buf = torch.zeros(want, dtype=torch.float32, device=device)
        elif self.used:
            self.buf[:self.used].zero_()
        self.off, self.used, self.need, self.active = 0, 0, 0, True

    def end_step(self):
        self.used, self.active = self.off, False

    def zeros(self, shape, device):
        n = 1
        for d in shape:
            n *= int(d)
        n_al = (n + 127) & ~127                              # 512-byte aligned slices, like the allocator's blocks (the kernels' atomics run by cache line)
        self.need += n_al
        if not self.active or self.buf is None or self.buf.device != torch.device(device) or self.off + n_al > self.buf.numel():
            return torch.zeros(shape, dtype=torch.float32, device=device)
        v = self.buf[self.off:self.off + n].view(shape)
        self.off += n_al
        self.used = self.off
        return v


grad_arena = _GradArena()


class _WgradQueue(object):
    """Weight gradients that wait for the end of the backward pass (PPOTrainer: `begin()` before `loss.backward()`, `flush()`
    right after it).  A tall-skinny layer's weight gradient over 10^4..10^5 rows is a launch of ramp and tail; nothing reads it before
    the optimiser step, so the backward only queues (x, dy, parameter) and ONE grouped launch per tile shape
    (catan_linear_wgrad_grouped) accumulates all of them - into whatever buffer the parameter's `.grad` is by then: the backward
    hands autograd a zero tensor for a parameter that has no `.grad` yet (autograd keeps it or copies it: either way `.grad` exists
    at the flush) and nothing for one that has (the flat bucket's view under several ranks, or an earlier use of a shared layer)."""

    def __init__(self):
        self.items, self.active, self.enabled = [], False, True

    def begin(self):
        self.items, self.active = [], bool(self.enabled)

    def take(self, x2, dy2, w, b, col0=None):
        """queue dW += dy2^T x2, db += column sums of dy2 for the leaf parameters w, b; -> what the backward returns for them.
        col0 is not None: the layer multiplied with the column slice w[:, col0 : col0 + I] of the parameter w (the heads' conditioning
        columns): the product goes straight into those columns of w's gradient and the slice's own gradient is None (no full-size
        zero tensor + copy + add per use)."""
        self.items.append((x2, dy2, w, b, col0))
        rw = None if (col0 is not None or w.grad is not None) else grad_zeros(tuple(w.shape), w.device)
        rb = None if (b is None or b.grad is not None) else grad_zeros(tuple(b.shape), b.device)
        return rw, rb

    @staticmethod
    def column_window(w):
        """(parameter, first column) when w is a column slice `param[:, c0:c1]` of a leaf fp32 parameter, else None"""
        base = getattr(w, "_base", None)
        if base is None or w.dim() != 2 or base.dim() != 2 or not (base.is_leaf and base.requires_grad and base.dtype == torch.float32 and base.is_contiguous()):
            return None
        if w.stride() != base.stride() or w.shape[0] != base.shape[0]:
            return None
        off = w.storage_offset() - base.storage_offset()
        return (base, off) if 0 <= off and off + w.shape[1] <= base.shape[1] else None

    def accepts(self, w, b):
        ok = lambda p: p.is_leaf and p.requires_grad and p.dtype == torch.float32 and p.is_contiguous()
        return self.active and ok(w) and (b is None or ok(b))

    def drop(self):
        """forget what was queued (the backward pass failed: its gradients are not wanted)"""
        self.items, self.active = [], False

    def flush(self):
        items, self.items, self.active = self.items, [], False
        if not items:
            return
        probs = (_lib.CatanWgradProblem * len(items))()
        for k, (x2, dy2, w, b, col0) in enumerate(items):
            if w.grad is None and col0 is not None:
                w.grad = torch.zeros_like(w)               # (no other use of the parameter produced a gradient in this pass)
            gw, gb = w.grad, (None if b is None else b.grad)
            if gw is None or gw.dtype != torch.float32 or not gw.is_contiguous() or (b is not None and (gb is None or gb.dtype != torch.float32 or not gb.is_contiguous())):
                raise RuntimeError("deferred weight gradient: the parameter has no fp32 contiguous .grad at the flush")
            probs[k] = _lib.CatanWgradProblem(x2.data_ptr(), dy2.data_ptr(), gw.data_ptr(), gb.data_ptr() if gb is not None else None,
                                              x2.shape[0], x2.shape[1], dy2.shape[1], 0 if col0 is None else w.shape[1], 0 if col0 is None else col0)
        _lib.check(_lib.lib().catan_linear_wgrad_grouped(probs, len(items), _stream()))


wgrad_queue = _WgradQueue()


def grad_zeros(shape, device):
    """a zeroed fp32 accumulator for a backward kernel (see _GradArena)"""
    return grad_arena.zeros(tuple(shape) if isinstance(shape, (tuple, list)) else (int(shape),), device)


class _SmallLayerNorm(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, b, eps, relu):
        D = x.shape[-1]
        x = _aligned(x)
        y = torch.empty_like(x)
        rows = x.numel() // D
        wf, bf = w.detach().float().contiguous(), b.detach().float().contiguous()
        _lib.check(_lib.lib().catan_layer_norm_fwd(_ptr(x), _ptr(wf), _ptr(bf), _ptr(y), rows, D, float(eps), int(relu),
                                                   int(x.dtype == torch.bfloat16), _stream()))
        ctx.save_for_backward(x, wf, bf)
        ctx.eps, ctx.relu = eps, relu
        return y

    @staticmethod
    def backward(ctx, dy):
        x, wf, bf = ctx.saved_tensors
        D = x.shape[-1]
        rows = x.numel() // D
        dy = _aligned(dy.to(x.dtype))
        dx = torch.empty_like(x)
        dwb = grad_zeros((2, D), x.device)      # both accumulators
        dw, db = dwb[0], dwb[1]
        _lib.check(_lib.lib().catan_layer_norm_bwd(_ptr(x), _ptr(wf), _ptr(bf), _ptr(dy), _ptr(dx), _ptr(dw), _ptr(db), rows, D,
                                                   float(ctx.eps), int(ctx.relu), int(x.dtype == torch.bfloat16), _stream()))
        return dx, dw, db, None, None


class _PreNorm(torch.autograd.Function):
    """(LayerNorm(x), x) for a pre-norm sub-layer `x + f(norm(x))`: x leaves through the function a second time, so the
    function is x's ONLY consumer and its backward forms x's whole gradient - the LayerNorm's plus the residual stream's - in
    the LayerNorm backward kernel (catan_layer_norm_bwd_res) instead of autograd adding two [rows, D] tensors afterwards."""

    @staticmethod
    def forward(ctx, x, w, b, eps):
        D = x.shape[-1]
        x = _aligned(x)
        y = torch.empty_like(x)
        rows = x.numel() // D
        wf, bf = w.detach().float().contiguous(), b.detach().float().contiguous()
        _lib.check(_lib.lib().catan_layer_norm_fwd(_ptr(x), _ptr(wf), _ptr(bf), _ptr(y), rows, D, float(eps), 0,
                                                   int(x.dtype == torch.bfloat16), _stream()))
        ctx.save_for_backward(x, wf, bf)
        ctx.eps = eps
        return y, x.view_as(x)

    @staticmethod
    def backward(ctx, dy, dres):
        x, wf, bf = ctx.saved_tensors
        D = x.shape[-1]
        rows = x.numel() // D
        dx = torch.empty_like(x)
        dwb = grad_zeros((2, D), x.device)
        L = _lib.lib()
        if dy is None:
            return dres, None, None, None
        dy = _aligned(dy.to(x.dtype))
        if dres is None:
            _lib.check(L.catan_layer_norm_bwd(_ptr(x), _ptr(wf), _ptr(bf), _ptr(dy), _ptr(dx), _ptr(dwb[0]), _ptr(dwb[1]), rows, D,
                                              float(ctx.eps), 0, int(x.dtype == torch.bfloat16), _stream()))
        else:
            dres = _aligned(dres.to(x.dtype))
            _lib.check(L.catan_layer_norm_bwd_res(_ptr(x), _ptr(wf), _ptr(bf), _ptr(dy), _ptr(dres), _ptr(dx), _ptr(dwb[0]), _ptr(dwb[1]),
                                                  rows, D, float(ctx.eps), 0, int(x.dtype == torch.bfloat16), _stream()))
        return dx, dwb[0], dwb[1], None


def pre_norm_supported(x, ln):
    return (x.is_cuda and x.dtype in (torch.float32, torch.bfloat16) and len(ln.normalized_shape) == 1
            and ln.normalized_shape[0] in (64, 128, 256, 512) and torch.is_grad_enabled() and x.requires_grad)


def pre_norm(x, ln):
    """-> (ln(x), x'): use x' as the residual stream (see _PreNorm)"""
    return _PreNorm.apply(x, ln.weight, ln.bias, ln.eps)


def small_layer_norm(x, ln, relu=False):
    """nn.LayerNorm `ln` (normalised dim in LN_WIDTHS) applied to CUDA tensor x (float32 / bfloat16), optional fused ReLU."""
    return _SmallLayerNorm.apply(x, ln.weight, ln.bias, ln.eps, relu)


def ln_supported(x, ln):
    return x.is_cuda and x.dtype in (torch.float32, torch.bfloat16) and len(ln.normalized_shape) == 1 and ln.normalized_shape[0] in LN_WIDTHS


class _WeightImages(object):
    """Every image of the fp32 master parameters that a training step under bf16 autocast reads - bf16 copies, transposed bf16
    copies, the fused tile encoder's packed blocks - kept in persistent buffers and refreshed by ONE kernel launch after the
    optimiser step (catan_weight_images).  An image is looked up by the parameter (view) it is made from; its source's version
    counter tells whether it is current, so a caller that never refreshes still gets correct values (one small copy per stale
    image, as before).  CATAN_WEIGHT_IMAGES=0 turns the registry off (every use casts, as autocast does)."""

    class Entry(object):
        __slots__ = ("src", "dst_view", "mode", "out", "version")

    def __init__(self):
        self.entries, self.by_key, self.dirty, self.table = [], {}, True, None
        self.enabled = os.environ.get("CATAN_WEIGHT_IMAGES", "1") != "0"

    MAX_ENTRIES = 8192       # a process that keeps building nets (a test session) starts over instead of growing the table for ever

    def clear(self):
        """forget every image (they are rebuilt on their next use); the sources of dropped nets are released with them"""
        global _TE_IMAGES
        self.entries, self.by_key, self.dirty, self.table = [], {}, True, None
        _TE_IMAGES = None

    def add(self, src2, dst_view, mode, out):
        """src2: 2-D fp32 view of a parameter; dst_view: a view of the image buffer indexed like src2 (any strides)"""
        if len(self.entries) >= self.MAX_ENTRIES:
            self.clear()
        e = _WeightImages.Entry()
        e.src, e.dst_view, e.mode, e.out, e.version = src2.detach(), dst_view, mode, out, -1
        self.entries.append(e)
        self.dirty = True
        return e

    @staticmethod
    def refresh_one(e):
        with torch.no_grad():
            e.dst_view.copy_(e.src.to(torch.bfloat16) if e.mode == 2 else e.src)
        e.version = e.src._version

    def image(self, w, transposed=False):
        """bf16 image of parameter (view) w - [O, I] as it is, or its transpose [I, O], contiguous - or None (not a CUDA fp32 tensor)"""
        if not self.enabled or not w.is_cuda or w.dtype != torch.float32 or w.dim() not in (1, 2) or (w.dim() == 1 and (transposed or not w.is_contiguous())):
            return None
        key = (w.data_ptr(), tuple(w.shape), tuple(w.stride()), transposed)
        e = self.by_key.get(key)
        if e is None:
            src2 = w.detach() if w.dim() == 2 else w.detach().view(1, -1)
            if transposed:
                out = torch.empty((w.shape[1], w.shape[0]), dtype=torch.bfloat16, device=w.device)
                dst_view = out.t()
            else:
                out = torch.empty(tuple(w.shape), dtype=torch.bfloat16, device=w.device)
                dst_view = out if w.dim() == 2 else out.view(1, -1)
            e = self.by_key[key] = self.add(src2, dst_view, 0, out)
        if e.version != w._version:
            self.refresh_one(e)
        return e.out

    def refresh_all(self):
        """one launch for every registered image (call after the optimiser step)"""
        if not self.enabled or not self.entries:
            return
        if self.dirty:
            import ctypes as C
            import numpy as np

            class Row(C.Structure):
                _fields_ = [("src", C.c_void_p), ("dst", C.c_void_p), ("rows", C.c_int32), ("cols", C.c_int32), ("s_r", C.c_int64), ("s_c", C.c_int64),
                            ("d_r", C.c_int64), ("d_c", C.c_int64), ("mode", C.c_int32), ("pad", C.c_int32)]
            assert C.sizeof(Row) == _lib.lib().catan_weight_image_bytes()
            rows = (Row * len(self.entries))()
            for r, e in zip(rows, self.entries):
                r.src, r.dst, r.rows, r.cols = e.src.data_ptr(), e.dst_view.data_ptr(), e.src.shape[0], e.src.shape[1]
                r.s_r, r.s_c, r.d_r, r.d_c, r.mode = e.src.stride(0), e.src.stride(1), e.dst_view.stride(0), e.dst_view.stride(1), e.mode
            host = torch.from_numpy(np.frombuffer(bytes(rows), dtype=np.uint8).copy())
            self.table = host.to(self.entries[0].src.device)
            self.dirty = False
        _lib.check(_lib.lib().catan_weight_images(_ptr(self.table), len(self.entries), _stream()))
        for e in self.entries:
            e.version = e.src._version


weight_images = _WeightImages()


def bf16_of(w):
    """w (fp32 parameter or view of one) in bf16: its registered image when there is one, else a cast"""
    img = weight_images.image(w) if w.dtype == torch.float32 else None
    return w.to(torch.bfloat16) if img is None else img


def bf16_t_of(w):
    """w^T in bf16, contiguous"""
    img = weight_images.image(w, transposed=True) if (w.dtype == torch.float32 and w.dim() == 2) else None
    return w.to(torch.bfloat16).t().contiguous() if img is None else img


class _LinearTallSkinny(torch.autograd.Function):
    """y = x @ w.T + b in bf16 (fp32 accumulate, library GEMM); backward: dx by the library, dw / db by the hand-written
    MFMA kernel k_wgrad (csrc/catan_nn.hip) which splits the huge row dimension over the grid."""

    @staticmethod
    def forward(ctx, x, w, b):
        with torch.autocast("cuda", enabled=False):
            xb = x.to(torch.bfloat16)
            wb = bf16_of(w)
            bb = None if b is None else bf16_of(b)
            # a wide input whose width is not a multiple of 8 (the opponents' 159 features) is zero-padded: the weight gradient
            # then takes the transposing-read kernel (614 400 x 159 -> 256: 685 us; x 160: 307 us) and the rows are 16-byte aligned
            # (x wider than w: the caller's rows are zero-padded already - policy.ObsParts - and only the weight gets its zero columns)
            extra = x.shape[-1] - w.shape[1]
            if extra < 0 or (extra > 0 and (x.shape[-1] % 8 or ctx.needs_input_grad[0])):
                raise ValueError("_LinearTallSkinny: input narrower than the weight, or a padded input that is not whole 16-byte pieces / needs a gradient")
            ctx.pad = pad = extra if extra > 0 else ((-x.shape[-1] % 8) if x.shape[-1] >= 64 else 0)
            if pad:
                xb, wb = (xb if extra > 0 else torch.nn.functional.pad(xb, (0, pad))), torch.nn.functional.pad(wb, (0, pad))
            y = _linear_rows(xb, wb, bb)
            if y is None:
                y = torch.nn.functional.linear(xb, wb, bb)
        ctx.save_for_backward(xb, wb, w, *(() if b is None else (b,)))
        ctx.has_bias = b is not None
        ctx.window = wgrad_queue.column_window(w) if (not pad and not w.is_leaf) else None
        return y

    @staticmethod
    def backward(ctx, dy):
        xb, wb, w = ctx.saved_tensors[:3]
        b = ctx.saved_tensors[3] if ctx.has_bias else None
        O, I = wb.shape
        dy2 = dy.reshape(-1, O).to(torch.bfloat16).contiguous()
        x2 = xb.reshape(-1, I).contiguous()
        dx = None
        if ctx.needs_input_grad[0]:
            dx = _linear_rows(dy2, wb.t().contiguous() if ctx.pad else bf16_t_of(w), None)          # dx[r][i] = sum_o dy[r][o] * w[o][i]
            dx = (dy2 @ wb if dx is None else dx).reshape(xb.shape)
        if not ctx.pad and not ((x2.data_ptr() | dy2.data_ptr()) & 15):
            if wgrad_queue.accepts(w, b):
                return (dx,) + wgrad_queue.take(x2, dy2, w, b)      # the weight gradient joins the grouped launch at the end of the backward pass
            if ctx.window is not None and wgrad_queue.accepts(ctx.window[0], b):
                return (dx,) + wgrad_queue.take(x2, dy2, ctx.window[0], b, col0=ctx.window[1])
        acc = grad_zeros((O * I + O,), dy.device)          # dw and db
        dw, db = acc[:O * I].view(O, I), (acc[O * I:] if ctx.has_bias else None)
        _lib.check(_lib.lib().catan_linear_wgrad(_ptr(x2), _ptr(dy2), _ptr(dw), _ptr(db), x2.shape[0], I, O, _stream()))
        if ctx.pad:
            dw = dw[:, :I - ctx.pad]
            dx = None if dx is None else dx[..., :I - ctx.pad]
        return dx, dw, db


ROWS_KERNEL_MIN_ROWS = 262144      # below this the library GEMM is as fast (measured: 65 536 x 128 x 128: 20 us vs 25 us)
ROWS_KERNEL_MIN_ROWS_SKINNY = 65536


MODE_NONE, MODE_RELU, MODE_ADD, MODE_RELU_MASK = 0, 1, 2, 3


def _linear_rows(x, w, b, aux=None, mode=MODE_NONE):
    """x [..., K] @ w[N, K].T (+ b) through k_linear_rows when the shape is in its range, else None.  mode / aux: the fused
    epilogue of catan_linear_rows_fused (ReLU, + residual, ReLU-backward mask)."""
    N, K = w.shape
    rows = x.numel() // K
    # (skinny products - at most 32 columns on one side - from 65 536 rows on: the library runs them at 0.3-0.7 TB/s, and its NN form, which the
    # input gradient of a narrow layer is, at a third of that: 204 800 x (128 -> 25): 68 us NN / 26 us NT / 19 us here; x (16 -> 25): 65 / 25 / 10 us;
    # x (128 -> 1): 44 / 43 / 11 us - profiles/r06_skinny_products.txt)
    min_rows = ROWS_KERNEL_MIN_ROWS_SKINNY if min(N, K) <= 32 else ROWS_KERNEL_MIN_ROWS
    if rows < min_rows or not _lib.lib().catan_linear_rows_supported(rows, K, N) or (mode != MODE_NONE and N % 8):
        return None
    x2, w2 = _aligned(x.reshape(rows, K)), _aligned(w)
    a2 = None if aux is None else _aligned(aux.reshape(rows, N))
    y = torch.empty((rows, N), dtype=torch.bfloat16, device=x.device)
    _lib.check(_lib.lib().catan_linear_rows_fused(_ptr(x2), _ptr(w2), _ptr(b.contiguous() if b is not None else None), _ptr(y), rows, K, N,
                                                  _ptr(a2), mode, _stream()))
    return y.reshape(x.shape[:-1] + (N,))


def _wgrad(x2, dy2, has_bias):
    O, I = dy2.shape[1], x2.shape[1]
    acc = grad_zeros((O * I + O,), dy2.device)             # dw and db
    dw, db = acc[:O * I].view(O, I), (acc[O * I:] if has_bias else None)
    _lib.check(_lib.lib().catan_linear_wgrad(_ptr(x2), _ptr(dy2), _ptr(dw), _ptr(db), x2.shape[0], I, O, _stream()))
    return dw, db


_WGRAD_BIG_WS = {}
WGRAD_BIG = True        # (tools/ab_step_switches.py)


def wgrad_big_supported(rows, in_features, out_features):
    return WGRAD_BIG and rows >= 16384 and in_features % 8 == 0 and in_features >= 256 and out_features % 128 == 0


def wgrad_big(x2, dy2, has_bias=True):
    """x2 [rows, I], dy2 [rows, O] bf16 -> (dW fp32 [O, I], db fp32 [O] or None) through catan_linear_wgrad_big: 128 x 128 output tiles, the
    row groups' partial tiles added in index order (deterministic)."""
    import ctypes as C
    rows, I, O = x2.shape[0], x2.shape[1], dy2.shape[1]
    L = _lib.lib()
    need = int(L.catan_wgrad_big_workspace_floats(rows, I, O))
    key = (x2.device, _stream().value)
    ws = _WGRAD_BIG_WS.get(key)
    if ws is None or ws.numel() < need:
        ws = _WGRAD_BIG_WS[key] = torch.empty(need, dtype=torch.float32, device=x2.device)
    out = torch.empty(O * I + O, dtype=torch.float32, device=x2.device)
    dw, db = out[:O * I].view(O, I), (out[O * I:] if has_bias else None)
    _lib.check(L.catan_linear_wgrad_big(_ptr(_aligned(x2)), _ptr(_aligned(dy2)), _ptr(dw), I, _ptr(db) if has_bias else None, _ptr(ws), rows, I, O, 0, _stream()))
    return dw, db


class _LinearBigWgrad(torch.autograd.Function):
    """y = x @ w.T + b (bf16, the library's GEMM) for a wide layer over many rows; backward: dX by the library, dW / db by k_wgrad_big (the
    library's split-K weight gradient of the trunk's [512 x R] . [R x 992] runs at 13 % of the MFMA peak).  w may be any tensor autograd
    tracks (the trunk passes its weight with five zero columns spliced in): its gradient comes back in fp32."""

    @staticmethod
    def forward(ctx, x, w, b):
        with torch.autocast("cuda", enabled=False):
            xb, wb = x.to(torch.bfloat16), w.to(torch.bfloat16)
            y = torch.nn.functional.linear(xb, wb, None if b is None else b.to(torch.bfloat16))
        ctx.save_for_backward(xb, wb)
        ctx.has_bias = b is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        xb, wb = ctx.saved_tensors
        with torch.autocast("cuda", enabled=False):
            dy2 = dy.reshape(-1, wb.shape[0]).to(torch.bfloat16).contiguous()
            dx = (dy2 @ wb).view(xb.shape) if ctx.needs_input_grad[0] else None
            dw, db = wgrad_big(xb.reshape(-1, xb.shape[-1]), dy2, ctx.has_bias)
        return dx, dw, db


def linear_big(x, w, b):
    return _LinearBigWgrad.apply(x, w, b)


def wgrad_grouped(pairs, has_bias=True):
    """[(x2 [rows_k, I_k], dy2 [rows_k, O_k]) bf16] -> [(dW_k fp32 [O_k, I_k], db_k [O_k])] through catan_linear_wgrad_grouped: problems of
    the same tile shape share launches (the heads' first layers on their row segments: 44 launches -> 2)."""
    import ctypes as C
    probs = (_lib.CatanWgradProblem * len(pairs))()
    keep, out = [], []
    for k, (x2, dy2) in enumerate(pairs):
        x2, dy2 = _aligned(x2), _aligned(dy2.to(torch.bfloat16))
        O, I = dy2.shape[1], x2.shape[1]
        acc = grad_zeros((O * I + O,), dy2.device)
        dw, db = acc[:O * I].view(O, I), (acc[O * I:] if has_bias else None)
        probs[k] = _lib.CatanWgradProblem(x2.data_ptr(), dy2.data_ptr(), dw.data_ptr(), db.data_ptr() if db is not None else None, x2.shape[0], I, O)
        keep.append((x2, dy2)); out.append((dw, db))
    _lib.check(_lib.lib().catan_linear_wgrad_grouped(probs, len(pairs), _stream()))
    return out


def wgrad_supported(rows, in_features, out_features):
    return bool(_lib.lib().catan_linear_wgrad_supported(rows, in_features, out_features))


def wgrad(x2, dy2):
    """dW [O, I] (fp32) and db [O] of y = x @ W.T + b from bf16 x2 [rows, I], dy2 [rows, O] (catan_linear_wgrad)"""
    return _wgrad(_aligned(x2), _aligned(dy2.to(torch.bfloat16)), True)


class _LinearResidual(torch.autograd.Function):
    """res + (x @ w.T + b): the out-projection of an encoder sub-layer with the residual add fused into the product's row
    stores (same values as the two separate ops: the product is rounded to bf16 before the add)."""

    @staticmethod
    def forward(ctx, x, w, b, res):
        xb, wb, bb, rb = x.to(torch.bfloat16), w.to(torch.bfloat16), b.to(torch.bfloat16), res.to(torch.bfloat16)
        y = _linear_rows(xb, wb, bb, rb, MODE_ADD)
        ctx.save_for_backward(xb, wb)
        return y

    @staticmethod
    def backward(ctx, dy):
        xb, wb = ctx.saved_tensors
        O, I = wb.shape
        dy2 = _aligned(dy.reshape(-1, O).to(torch.bfloat16))
        dx = _linear_rows(dy2, wb.t().contiguous(), None).reshape(xb.shape) if ctx.needs_input_grad[0] else None
        dw, db = _wgrad(_aligned(xb.reshape(-1, I)), dy2, True)
        return dx, dw, db, (dy if ctx.needs_input_grad[3] else None)


class _FFNResidual(torch.autograd.Function):
    """res + linear2(relu(linear1(x))) (the pointwise net of an encoder layer, reference pointwise_feedforward) as two
    row-kernel launches: ReLU fused into the first product, the residual add into the second; the backward masks the hidden
    gradient inside the dX product (aux = the ReLU output) - no elementwise pass touches the [rows, 2 dim] hidden tensor."""

    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2, res):
        bf = torch.bfloat16
        xb, w1b, w2b = x.to(bf), w1.to(bf), w2.to(bf)
        h = _linear_rows(xb, w1b, b1.to(bf), None, MODE_RELU)
        y = _linear_rows(h, w2b, b2.to(bf), res.to(bf), MODE_ADD)
        ctx.save_for_backward(xb, h, w1b, w2b)
        return y

    @staticmethod
    def backward(ctx, dy):
        xb, h, w1b, w2b = ctx.saved_tensors
        D, F2 = w2b.shape                                          # w2 [dim][hidden]
        dy2 = _aligned(dy.reshape(-1, D).to(torch.bfloat16))
        h2 = h.reshape(-1, F2)
        dh = _linear_rows(dy2, w2b.t().contiguous(), None, h2, MODE_RELU_MASK)           # (dy @ w2) where h > 0
        dw2, db2 = _wgrad(_aligned(h2), dy2, True)
        dw1, db1 = _wgrad(_aligned(xb.reshape(-1, D)), dh, True)
        dx = _linear_rows(dh, w1b.t().contiguous(), None).reshape(xb.shape) if ctx.needs_input_grad[0] else None
        return dx, dw1, db1, dw2, db2, (dy if ctx.needs_input_grad[5] else None)


def fused_sublayer_supported(x, dim, hidden=None):
    """training / inference on the GPU in bf16 with enough rows for the row kernels, widths they are built for"""
    if not x.is_cuda or x.shape[-1] != dim or dim % 8 or x.numel() // dim < ROWS_KERNEL_MIN_ROWS:
        return False
    if not (x.dtype == torch.bfloat16 or (torch.is_autocast_enabled() and torch.get_autocast_dtype("cuda") == torch.bfloat16)):
        return False
    L = _lib.lib()
    rows = x.numel() // dim
    ok = L.catan_linear_rows_supported(rows, dim, dim) and L.catan_linear_wgrad_supported(rows, dim, dim)
    if hidden is not None:
        ok = ok and hidden % 8 == 0 and L.catan_linear_rows_supported(rows, dim, hidden) and L.catan_linear_rows_supported(rows, hidden, dim) \
            and L.catan_linear_wgrad_supported(rows, dim, hidden) and L.catan_linear_wgrad_supported(rows, hidden, dim)
    return bool(ok)


def linear_residual(x, w, b, res):
    with torch.autocast("cuda", enabled=False):
        return _LinearResidual.apply(x, w, b, res)


def ffn_residual(x, w1, b1, w2, b2, res):
    with torch.autocast("cuda", enabled=False):
        return _FFNResidual.apply(x, w1, b1, w2, b2, res)


def linear_inference(x, w, b=None):
    """Forward only (no autograd), bf16: the tall-skinny kernel when the shape is in its range, else None."""
    with torch.autocast("cuda", enabled=False):
        return _linear_rows(x.to(torch.bfloat16), w.to(torch.bfloat16), None if b is None else b.to(torch.bfloat16))


def linear_supported(x, w):
    """bf16 compute on the GPU (autocast or bf16 input), widths the kernel is built for, and enough rows to be worth it."""
    if not x.is_cuda or w.dim() != 2:
        return False
    if not (x.dtype == torch.bfloat16 or (torch.is_autocast_enabled() and torch.get_autocast_dtype('cuda') == torch.bfloat16)):
        return False
    O, I = w.shape
    rows = x.numel() // x.shape[-1]        # (x may be wider than w: rows already zero-padded to whole 16-byte pieces, _LinearTallSkinny)
    return rows >= 4096 and bool(_lib.lib().catan_linear_wgrad_supported(rows, I, O))


def linear(x, w, b=None):
    """Linear layer for tall-skinny shapes (see linear_supported); same values as F.linear under bf16 autocast."""
    return _LinearTallSkinny.apply(x, w, b)


def use_tuned_gemms():
    """Library GEMMs with the solutions PyTorch TunableOp found for this net's shapes on gfx950 (tools/tune_gemms.py ->
    tunableop_gfx950.csv: rocBLAS / hipBLASLt solution indices per (transposes, m, n, k) at the rollout, minibatch and value-
    chunk widths).  Tuning itself stays off: shapes that are not in the file use the library default.  The file carries
    the library versions it was made with; TunableOp ignores it when they differ."""
    import os
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tunableop_gfx950.csv")
    if not (torch.cuda.is_available() and os.path.exists(path)):
        return False
    import torch.cuda.tunable as tun
    tun.enable(True)
    tun.tuning_enable(False)
    tun.set_filename(os.path.join(os.environ.get("TMPDIR", "/tmp"), "catan_tunableop_unused.csv"))   # never write into the package
    return bool(tun.read_file(path))


class _LSTMCell(torch.autograd.Function):
    """One LSTM step between its two GEMMs (csrc/catan_nn.hip k_lstm_cell_fwd / _bwd)."""

    @staticmethod
    def forward(ctx, gx, gh, c_prev, mask):
        n, L = c_prev.shape
        gx, gh, c_prev = _aligned(gx), _aligned(gh), _aligned(c_prev)
        mask = None if mask is None else mask.contiguous()
        h = torch.empty((n, L), dtype=torch.float32, device=gx.device)
        c = torch.empty_like(h)
        _lib.check(_lib.lib().catan_lstm_cell_fwd(_ptr(gx), _ptr(gh), _ptr(c_prev), _ptr(mask), _ptr(h), _ptr(c), n, L,
                                                  int(gx.dtype == torch.bfloat16), _stream()))
        ctx.save_for_backward(gx, gh, c_prev, mask)
        return h, c

    @staticmethod
    def backward(ctx, dh, dc):
        gx, gh, c_prev, mask = ctx.saved_tensors
        n, L = c_prev.shape
        dh, dc = _aligned(dh.float()), _aligned(dc.float())
        dg = torch.empty_like(gx)
        dcp = torch.empty_like(c_prev)
        _lib.check(_lib.lib().catan_lstm_cell_bwd(_ptr(gx), _ptr(gh), _ptr(c_prev), _ptr(mask), _ptr(dh), _ptr(dc), _ptr(dg), _ptr(dcp),
                                                  n, L, int(gx.dtype == torch.bfloat16), _stream()))
        return dg, dg, dcp, None


def lstm_cell_supported(gx, c_prev):
    return gx.is_cuda and gx.dtype in (torch.float32, torch.bfloat16) and c_prev.dtype == torch.float32 and c_prev.shape[-1] % 4 == 0


def lstm_cell(gx, gh, c_prev, mask=None):
    """gx, gh [n, 4L] (same dtype, fp32 or bf16; gate order i, f, g, o), c_prev fp32 [n, L], mask fp32 [n] or None
    -> (h, c) fp32 [n, L] with c = sigmoid(f) * (c_prev * mask) + sigmoid(i) * tanh(g), h = sigmoid(o) * tanh(c)."""
    assert gx.shape == gh.shape and gx.dtype == gh.dtype and gx.shape[-1] == 4 * c_prev.shape[-1]
    return _LSTMCell.apply(gx, gh, c_prev, None if mask is None else mask.float().reshape(-1))


def _pad2(w, rows, cols):
    out = torch.zeros((rows, cols), dtype=torch.bfloat16, device=w.device)
    out[:w.shape[0], :w.shape[1]] = w.to(torch.bfloat16)
    return out.reshape(-1)


def _pad1(v, n):
    out = torch.zeros((n,), dtype=torch.float32, device=v.device)
    out[:v.shape[0]] = v.float()
    return out


def tile_encoder_pack(te):
    """The tile encoder's parameters in the layout of catan_tile_encoder_fwd (include/catan_hip.h), cached on the module and
    re-packed when a parameter changed (optimiser step, inference-copy refresh)."""
    params = list(te.parameters())
    stamp = (sum(p._version for p in params), params[0].device, params[0].data_ptr())
    cache = getattr(te, "_fused_pack", None)
    if cache is not None and cache[0] == stamp:
        return cache[1], cache[2]
    def lb(bias, n):                                           # a Linear bias as bf16 autocast hands it to the GEMM (rounded once)
        return _pad1(bias.to(torch.bfloat16), n)
    with torch.no_grad():
        w = [_pad2(te.first_layer.weight, 64, 64)]
        v = [lb(te.first_layer.bias, 64), _pad1(te.norm_2.weight, 64), _pad1(te.norm_2.bias, 64)]
        for layer in te.encoder_layers:
            mha, ffn = layer.multi_headed_attention, layer.pointwise_net
            w += [_pad2(torch.cat([n.weight for n in mha.qkv_nets], 0), 192, 64), _pad2(mha.out_proj_net.weight, 64, 64),
                  _pad2(ffn.linear1.weight, 128, 64), _pad2(ffn.linear2.weight, 64, 128)]
            v += [_pad1(layer.sublayers[0].norm.weight, 64), _pad1(layer.sublayers[0].norm.bias, 64),
                  lb(torch.cat([n.bias for n in mha.qkv_nets], 0), 192), lb(mha.out_proj_net.bias, 64),
                  _pad1(layer.sublayers[1].norm.weight, 64), _pad1(layer.sublayers[1].norm.bias, 64),
                  lb(ffn.linear1.bias, 128), lb(ffn.linear2.bias, 64)]
        w.append(_pad2(te.out_proj.weight, 32, 64))
        v += [lb(te.out_proj.bias, 32), _pad1(te.norm.weight, 32), _pad1(te.norm.bias, 32)]
        wts, vecs = torch.cat(w).contiguous(), torch.cat(v).contiguous()
    L = _lib.lib()
    assert wts.numel() == L.catan_tile_encoder_weight_elems() and vecs.numel() == L.catan_tile_encoder_vec_elems()
    if cache is not None and cache[1].device == wts.device:
        # re-pack IN PLACE: a captured hipGraph of the forward (GraphedAct) holds these buffers' addresses
        cache[1].copy_(wts); cache[2].copy_(vecs)
        wts, vecs = cache[1], cache[2]
    te._fused_pack = (stamp, wts, vecs)
    return wts, vecs


def tile_encoder_supported(te, tiles):
    """inference (no autograd), GPU, bf16 compute, the reference's sizes (19 tiles x 60 features, 64 wide, 4 heads, 2 layers, 25 out)"""
    if torch.is_grad_enabled() or not tiles.is_cuda or tiles.dim() != 3 or tuple(tiles.shape[1:]) != (19, 60):
        return False
    if not (tiles.dtype == torch.bfloat16 or (torch.is_autocast_enabled() and torch.get_autocast_dtype("cuda") == torch.bfloat16)):
        return False
    return (te.first_layer.weight.shape == (64, 60) and len(te.encoder_layers) == 2 and te.out_proj.weight.shape == (25, 64)
            and te.encoder_layers[0].multi_headed_attention.heads == 4 and te.encoder_layers[0].pointwise_net.linear1.weight.shape == (128, 64))


def tile_encoder_forward(te, tiles):
    """tiles [B, 19, 60] -> bf16 [B, 475]: the whole tile encoder in one kernel (k_tile_encoder_fwd)"""
    wts, vecs = tile_encoder_pack(te)
    x = _aligned(tiles.to(torch.bfloat16))
    out = torch.empty((x.shape[0], 19 * 25), dtype=torch.bfloat16, device=x.device)
    _lib.check(_lib.lib().catan_tile_encoder_fwd(_ptr(x), _ptr(wts), _ptr(vecs), _ptr(out), x.shape[0], _stream()))
    return out


def gather_rows(src, idx, out=None):
    """src[idx] for a row-major tensor whose rows are contiguous (any row pitch: a column window of a wider matrix is fine) - into
    `out` (same row shape, contiguous rows, any pitch) or a new contiguous tensor.  catan_gather_rows on the GPU for rows of an
    even number of bytes (the bf16 rollout rows are only 2-byte aligned: the generic indexing kernel moves them per element)."""
    row_shape = src.shape[1:]
    cols = 1
    for d in row_shape:
        cols *= d
    ok = (src.is_cuda and idx.dtype == torch.int64 and idx.dim() == 1 and idx.numel() > 0 and (cols * src.element_size()) % 2 == 0
          and src[0].is_contiguous() and (out is None or (out[0].is_contiguous() and out.shape[1:] == row_shape and out.dtype == src.dtype)))
    if not ok:
        if out is None:
            return src[idx]
        out.copy_(src[idx])
        return out
    if out is None:
        out = torch.empty((idx.numel(),) + tuple(row_shape), dtype=src.dtype, device=src.device)
    es = src.element_size()
    _lib.check(_lib.lib().catan_gather_rows(_ptr(src), src.stride(0) * es, _ptr(idx.contiguous()), idx.numel(), _ptr(out), out.stride(0) * es, cols * es, _stream()))
    return out


class _ExpandRows(torch.autograd.Function):
    """out[j] = src[inv[j]] (a per-board result spread to the rows that show the board); backward: the rows' gradients summed per
    board in fp32 over the board's rows (order / start: the rows sorted by board and where each board's run starts) - autograd's
    indexing backward sorts the 204 800 indices again in every step and accumulates through index_put (0.87 ms)."""

    @staticmethod
    def forward(ctx, src, inv, order, start):
        src = src.contiguous()
        out = torch.empty((inv.numel(), src.shape[1]), dtype=src.dtype, device=src.device)
        _lib.check(_lib.lib().catan_expand_rows(_ptr(src), _ptr(inv), inv.numel(), _ptr(out), src.shape[1] * src.element_size(), _stream()))
        ctx.save_for_backward(order, start)
        ctx.U = src.shape[0]
        return out

    @staticmethod
    def backward(ctx, dy):
        order, start = ctx.saved_tensors
        if dy.stride(1) != 1 or (dy.stride(0) * 2) % 16 or dy.data_ptr() % 16:          # (a column window of a wider gradient is taken in place)
            dy = dy.contiguous()
        dsrc = torch.empty((ctx.U, dy.shape[1]), dtype=dy.dtype, device=dy.device)
        _lib.check(_lib.lib().catan_segment_sum_rows(_ptr(dy), dy.stride(0) * 2, _ptr(order), _ptr(start), ctx.U, _ptr(dsrc), dy.shape[1] * 2, _stream()))
        return dsrc, None, None, None


def expand_rows(src, inv, order=None, start=None):
    """src [U, W] -> [len(inv), W]; with (order, start) of the rows' boards and bf16 rows of whole 16-byte pieces on the GPU: the
    kernels above, else plain indexing"""
    if (order is not None and start is not None and src.is_cuda and src.dim() == 2 and src.dtype == torch.bfloat16 and (src.shape[1] * 2) % 16 == 0
            and inv.dtype == torch.int64 and start.numel() == src.shape[0] + 1):
        return _ExpandRows.apply(src, inv.contiguous(), order.contiguous(), start.contiguous())
    return src[inv]


class _ConcatRows(torch.autograd.Function):
    """torch.cat(parts, -1) of 2-D bf16 tensors whose rows are whole 16-byte pieces, by catan_concat_rows; backward: the column windows of
    the gradient (views, as cat's own backward hands out)."""

    @staticmethod
    def forward(ctx, *parts):
        parts = [p.contiguous() for p in parts]
        ctx.widths = [p.shape[1] for p in parts]
        rows, W = parts[0].shape[0], sum(ctx.widths)
        out = torch.empty((rows, W), dtype=parts[0].dtype, device=parts[0].device)
        srcs = (C.c_void_p * len(parts))(*[p.data_ptr() for p in parts])
        rb = (C.c_int64 * len(parts))(*[w * 2 for w in ctx.widths])
        _lib.check(_lib.lib().catan_concat_rows(C.cast(srcs, C.c_void_p), C.cast(rb, C.c_void_p), len(parts), _ptr(out), W * 2, rows, _stream()))
        return out

    @staticmethod
    def backward(ctx, dy):
        out, c = [], 0
        for w in ctx.widths:
            out.append(dy[:, c:c + w]); c += w
        return tuple(out)


CONCAT_ROWS = True            # (A/B switch)


def concat_rows(parts):
    """torch.cat(parts, -1); on the GPU for 2..4 bf16 matrices of equal row count whose rows are whole 16-byte pieces: one kernel at the
    HBM rate"""
    if (CONCAT_ROWS and 2 <= len(parts) <= 4 and all(p.is_cuda and p.dim() == 2 and p.dtype == torch.bfloat16 and (p.shape[1] * 2) % 16 == 0
                                                      and p.shape[0] == parts[0].shape[0] for p in parts) and parts[0].shape[0] > 0):
        return _ConcatRows.apply(*parts)
    return torch.cat(parts, -1)


def _scatter_ranges(dy, perm, ranges, U, add0=None, add1=None):
    if dy.stride(1) != 1 or (dy.stride(0) * 2) % 16 or dy.data_ptr() % 16:
        dy = dy.contiguous()
    adds = [None if a is None else a.to(dy.dtype).contiguous() for a in (add0, add1)]
    dsrc = torch.empty((U, dy.shape[1]), dtype=dy.dtype, device=dy.device)
    flat = (C.c_int64 * (3 * len(ranges)))(*[int(v) for r in ranges for v in r])
    _lib.check(_lib.lib().catan_scatter_rows_ranges(_ptr(dy), dy.stride(0) * 2, _ptr(perm), U, C.cast(flat, C.c_void_p), len(ranges),
                                                     None if adds[0] is None else _ptr(adds[0]), None if adds[1] is None else _ptr(adds[1]), _ptr(dsrc),
                                                     dy.shape[1] * 2, _stream()))
    return dsrc


class _GatherRanges(torch.autograd.Function):
    """out = src[idx] where idx is the concatenation of the ranges perm[a:b] (`ranges`: host list of (a, b, off), off = where the range
    starts in idx); backward: catan_scatter_rows_ranges - one pass over dy instead of index_put's sort + accumulate."""

    @staticmethod
    def forward(ctx, src, perm, idx, ranges):
        ctx.save_for_backward(perm)
        ctx.ranges, ctx.U = ranges, src.shape[0]
        return gather_rows(src, idx)

    @staticmethod
    def backward(ctx, dy):
        perm, = ctx.saved_tensors
        return _scatter_ranges(dy, perm, ctx.ranges, ctx.U), None, None, None


class _FanOutGatherRanges(torch.autograd.Function):
    """(src, src, src[idx]): the source tensor leaves through the function for its two other consumers as well, so the function is
    its ONLY consumer and forms its whole gradient - the gathered rows' sums plus the two consumers' gradients - in the one pass of
    catan_scatter_rows_ranges instead of autograd adding full-size tensors twice afterwards."""

    @staticmethod
    def forward(ctx, src, perm, idx, ranges):
        ctx.save_for_backward(perm)
        ctx.ranges, ctx.U, ctx.W = ranges, src.shape[0], src.shape[1]
        return src.view_as(src), src.view_as(src), gather_rows(src, idx)

    @staticmethod
    def backward(ctx, g0, g1, dy):
        perm, = ctx.saved_tensors
        if dy is None:
            dy = torch.zeros((1, ctx.W), dtype=(g0 if g0 is not None else g1).dtype, device=perm.device)
            return _scatter_ranges(dy, perm, (), ctx.U, g0, g1), None, None, None
        return _scatter_ranges(dy, perm, ctx.ranges, ctx.U, g0, g1), None, None, None


GATHER_RANGES = True          # (A/B switches)
FANOUT_GATHER = True


def _gather_ranges_ok(src, perm, ranges):
    return (GATHER_RANGES and src.is_cuda and src.dim() == 2 and src.dtype == torch.bfloat16 and (src.shape[1] * 2) % 16 == 0 and src.is_contiguous()
            and perm.dtype == torch.int64 and perm.numel() == src.shape[0] and 0 < len(ranges) <= 16 and torch.is_grad_enabled() and src.requires_grad)


def gather_ranges(src, perm, idx, ranges):
    """src[idx] for idx = cat(perm[a:b] for (a, b, off) in ranges) with the ranges-aware backward on the GPU (bf16 rows of whole
    16-byte pieces, perm a permutation of all of src's rows, at most 16 ranges); plain indexing otherwise"""
    if _gather_ranges_ok(src, perm, ranges):
        return _GatherRanges.apply(src, perm.contiguous(), idx, tuple(ranges))
    return src[idx]


def fanout_gather_ranges(src, perm, idx, ranges):
    """-> (src, src, src[idx]) for a tensor with exactly these three consumers (see _FanOutGatherRanges); (src, src, gather) otherwise"""
    if FANOUT_GATHER and _gather_ranges_ok(src, perm, ranges):
        return _FanOutGatherRanges.apply(src, perm.contiguous(), idx, tuple(ranges))
    return src, src, gather_ranges(src, perm, idx, ranges)


# ---------------------------------------------------------------------------------------------------------------------
# The tile encoder's TRAINING forward as the one fused kernel (k_tile_encoder_fwd<SAVE>): the minibatch steps of a PPO update
# spent 5.5 of their 41 ms in the encoder's forward as ~20 kernels that each stream a [boards x 19, 64..192] activation tensor
# through HBM; the fused kernel keeps a board on chip and only WRITES what the backward kernels read.  The backward is the
# chain autograd ran through the unfused sub-layers (same kernels, same order), spelled out.
_TE_SAVES = (("tiles64", 64), ("a0", 64), ("xin0", 64), ("xin1", 64), ("n1_0", 64), ("n1_1", 64), ("qkv0", 192), ("qkv1", 192), ("o0", 64), ("o1", 64),
             ("xmid0", 64), ("xmid1", 64), ("n2_0", 64), ("n2_1", 64), ("h0", 128), ("h1", 128), ("xfin", 64), ("p", 25))     # catan_te_saves_t's order


def _te_params(te):
    ps = [te.first_layer.weight, te.first_layer.bias, te.norm_2.weight, te.norm_2.bias]
    for layer in te.encoder_layers:
        mha, ffn = layer.multi_headed_attention, layer.pointwise_net
        ps += [layer.sublayers[0].norm.weight, layer.sublayers[0].norm.bias]
        for n in mha.qkv_nets:
            ps += [n.weight, n.bias]
        ps += [mha.out_proj_net.weight, mha.out_proj_net.bias, layer.sublayers[1].norm.weight, layer.sublayers[1].norm.bias,
               ffn.linear1.weight, ffn.linear1.bias, ffn.linear2.weight, ffn.linear2.bias]
    return ps + [te.out_proj.weight, te.out_proj.bias, te.norm.weight, te.norm.bias]


def _rows_product(x2, wt, aux=None, mode=MODE_NONE):
    """x2 [rows, K] @ wt[N, K].T (bf16), optional ReLU-backward mask: the row kernel, or the library for shapes outside its range"""
    y = _linear_rows(x2, wt, None, aux, mode)
    if y is None:
        y = torch.nn.functional.linear(x2, wt)
        if mode == MODE_RELU_MASK:
            y = y * (aux > 0).to(y.dtype)
    return y


def _ln_backward(x, w, b, dy, eps, relu, dres=None):
    """-> (dx, dw, db) of LayerNorm (+ ReLU) over the last dim of bf16 x [rows, D]; dres: a second gradient of x added in"""
    rows, D = x.shape
    dx = torch.empty_like(x)
    dwb = grad_zeros((2, D), x.device)
    wf, bf = w.detach().float().contiguous(), b.detach().float().contiguous()
    L = _lib.lib()
    if dres is None:
        _lib.check(L.catan_layer_norm_bwd(_ptr(x), _ptr(wf), _ptr(bf), _ptr(dy), _ptr(dx), _ptr(dwb[0]), _ptr(dwb[1]), rows, D, float(eps), int(relu), 1, _stream()))
    else:
        _lib.check(L.catan_layer_norm_bwd_res(_ptr(x), _ptr(wf), _ptr(bf), _ptr(dy), _ptr(dres), _ptr(dx), _ptr(dwb[0]), _ptr(dwb[1]), rows, D, float(eps),
                                              int(relu), 1, _stream()))
    return dx, dwb[0], dwb[1]


class _TeImages(object):
    """The training-side images of one tile encoder's parameters in weight_images: the packed weight / vector blocks of
    catan_tile_encoder_fwd (tile_encoder_pack's layout) and the transposed bf16 weights its backward chain multiplies by."""

    def __init__(self, te):
        L = _lib.lib()
        dev = te.first_layer.weight.device
        self.wts = torch.zeros((L.catan_tile_encoder_weight_elems(),), dtype=torch.bfloat16, device=dev)
        self.vecs = torch.zeros((L.catan_tile_encoder_vec_elems(),), dtype=torch.float32, device=dev)
        self.entries = []
        wo, vo = [0], [0]

        def W(param, rows, cols):                    # a weight block [rows][cols] of the pack, the parameter in its top-left corner
            blk = self.wts[wo[0]:wo[0] + rows * cols].view(rows, cols)
            self.entries.append(weight_images.add(param, blk[:param.shape[0], :param.shape[1]], 0, None))
            wo[0] += rows * cols

        def Wrows(params, rows, cols):               # ... several parameters stacked along the rows (Q, K, V)
            blk = self.wts[wo[0]:wo[0] + rows * cols].view(rows, cols)
            r = 0
            for prm in params:
                self.entries.append(weight_images.add(prm, blk[r:r + prm.shape[0], :prm.shape[1]], 0, None))
                r += prm.shape[0]
            wo[0] += rows * cols

        def V(params, n, mode):                      # a vector block of n floats; mode 2: a Linear bias as bf16 autocast rounds it
            o = 0
            for prm in params:
                self.entries.append(weight_images.add(prm.view(1, -1), self.vecs[vo[0] + o:vo[0] + o + prm.numel()].view(1, -1), mode, None))
                o += prm.numel()
            vo[0] += n

        def T(params):                               # transposed bf16 [in][sum of outs]
            outs = sum(prm.shape[0] for prm in params)
            buf = torch.empty((params[0].shape[1], outs), dtype=torch.bfloat16, device=dev)
            c = 0
            for prm in params:
                self.entries.append(weight_images.add(prm, buf[:, c:c + prm.shape[0]].t(), 0, None))
                c += prm.shape[0]
            return buf

        W(te.first_layer.weight, 64, 64)
        V([te.first_layer.bias], 64, 2); V([te.norm_2.weight], 64, 1); V([te.norm_2.bias], 64, 1)
        self.w2t, self.w1t, self.wot, self.wqt = [], [], [], []
        for layer in te.encoder_layers:
            mha, ffn = layer.multi_headed_attention, layer.pointwise_net
            Wrows([n.weight for n in mha.qkv_nets], 192, 64); W(mha.out_proj_net.weight, 64, 64); W(ffn.linear1.weight, 128, 64); W(ffn.linear2.weight, 64, 128)
            V([layer.sublayers[0].norm.weight], 64, 1); V([layer.sublayers[0].norm.bias], 64, 1)
            V([n.bias for n in mha.qkv_nets], 192, 2); V([mha.out_proj_net.bias], 64, 2)
            V([layer.sublayers[1].norm.weight], 64, 1); V([layer.sublayers[1].norm.bias], 64, 1)
            V([ffn.linear1.bias], 128, 2); V([ffn.linear2.bias], 64, 2)
            self.w2t.append(T([ffn.linear2.weight])); self.w1t.append(T([ffn.linear1.weight]))
            self.wot.append(T([mha.out_proj_net.weight])); self.wqt.append(T([n.weight for n in mha.qkv_nets]))
        W(te.out_proj.weight, 32, 64)
        V([te.out_proj.bias], 32, 2); V([te.norm.weight], 32, 1); V([te.norm.bias], 32, 1)
        self.wpt = T([te.out_proj.weight])
        assert wo[0] == self.wts.numel() and vo[0] == self.vecs.numel()
        self.stamp = (te.first_layer.weight.data_ptr(), dev)

    def current(self):
        for e in self.entries:
            if e.version != e.src._version:
                weight_images.refresh_one(e)
        return self


_TE_IMAGES = None


def _te_images(te):
    global _TE_IMAGES
    if _TE_IMAGES is None:
        import weakref
        _TE_IMAGES = weakref.WeakKeyDictionary()              # (not on the module: a deepcopy - inference_copy - must not carry them along)
    im = _TE_IMAGES.get(te)
    if im is None or im.stamp != (te.first_layer.weight.data_ptr(), te.first_layer.weight.device):
        im = _TeImages(te)
        if _TE_IMAGES is None:                                # (the registry started over while the images were being registered)
            import weakref
            _TE_IMAGES = weakref.WeakKeyDictionary()
        _TE_IMAGES[te] = im
    return im.current()


class _TeWorkspace(object):
    """the one persistent activation workspace of _TileEncoderTrain (grow-only), leased to one forward at a time"""
    buf, busy = None, False

    class Lease(object):
        def __init__(self, buf):
            self.buf, self.live = buf, True

        def release(self):
            if self.live:
                self.live = False
                _TeWorkspace.busy = False

        __del__ = release                       # (a forward whose graph is dropped without a backward gives the workspace back too)

    @classmethod
    def lease(cls, n, device):
        if cls.busy or os.environ.get("CATAN_TE_WORKSPACE", "1") == "0":
            return None
        if cls.buf is None or cls.buf.device != torch.device(device) or cls.buf.numel() < n:
            cls.buf = None                      # (free the old one first)
            cls.buf = torch.empty((int(n * 1.02) + 1024,), dtype=torch.bfloat16, device=device)
        cls.busy = True
        return cls.Lease(cls.buf)


def _te_backward_fused_w():
    """The tile encoder's backward runs k_ffn_bwd_w / k_qkv_bwd_w (the default; the toggles select the older chains for tests)."""
    return os.environ.get("CATAN_TE_BWD_UNFUSED") != "1" and os.environ.get("CATAN_TE_BWD_W", "1") == "1"


class _TileEncoderTrain(torch.autograd.Function):
    @staticmethod
    def forward(ctx, tiles, te, out_cols, *params):
        import ctypes as C
        if weight_images.enabled:
            im = _te_images(te)
            wts, vecs = im.wts, im.vecs
        else:
            wts, vecs = tile_encoder_pack(te)
        ctx.te = te
        x = _aligned(tiles.detach().to(torch.bfloat16))
        B = x.shape[0]
        T = B * 19
        # 3 KB per token = 11 GB at a minibatch's 180 000 boards, and the board count differs from step to step: taken from the
        # caching allocator every step, the slightly different sizes fragment it (reserved memory grew from 100 to 190 GB in three
        # updates).  ONE workspace is kept instead and leased to the forward whose backward has not run yet; a second forward
        # in flight (gradient accumulation) gets a fresh buffer as before.
        # The LayerNorm outputs n1 / n2 are NOT stored when the backward runs the one-pass kernels with the weight gradients: k_qkv_bwd_w
        # and k_ffn_bwd_w recompute them from the LayerNorm inputs they read anyway (at no cost: 754 vs 769 us and 1.35 vs 1.37 ms at
        # 3.9 M rows), and the forward writes 512 B per token less (3.04 -> 2.76 ms at 204 800 boards).  CATAN_TE_RECOMPUTE_N=1: only n1
        # recomputed, 0: both stored and read (tools/bench_te_n_recompute.py, the tests).
        level = int(os.environ.get("CATAN_TE_RECOMPUTE_N", "2")) if _te_backward_fused_w() else 0
        drop = ("n1_", "n2_")[:level]
        # CATAN_TE_RECOMPUTE_H=1: the FFN's hidden activation h (the widest one: 256 of the 1 024 bytes per token and layer) is not stored either:
        # k_ffn_bwd_w<., true, true> recomputes it from the recomputed n2 - one more 16 x 64 x 128 product per wave and stage.  OFF by default:
        # measured at config 3's minibatch (tools/ab_step_switches.py, profiles/r05_ab_recompute_h.txt) the step is 22.12 ms with h stored
        # and 22.25 ms with h recomputed - the pass is not bound by the bytes it reads, the product costs what the 9.7 KB per board save
        ctx.recompute_h = level == 2 and os.environ.get("CATAN_TE_RECOMPUTE_H", "0") == "1" and os.environ.get("CATAN_TE_BWD_OP", "1") == "1"
        if ctx.recompute_h:
            drop = drop + ("h",)
            ctx.packed = (wts, vecs)
        names = [(n, w) for n, w in _TE_SAVES if not (drop and n.startswith(drop))]
        need = T * sum(w for _, w in names)
        lease = _TeWorkspace.lease(need, x.device)
        ctx.lease = lease
        buf = lease.buf if lease is not None else torch.empty((need,), dtype=torch.bfloat16, device=x.device)
        saves, off = [], 0
        for _, w in names:
            saves.append(buf[off:off + T * w].view(T, w))
            off += T * w
        by_name = dict(zip([n for n, _ in names], saves))
        ptrs = (C.c_void_p * len(_TE_SAVES))(*[by_name[n].data_ptr() if n in by_name else None for n, _ in _TE_SAVES])
        ctx.save_names = [n for n, _ in names]
        # out_cols > 475: board rows padded with zero columns to whole 16-byte pieces (nn_kernels.expand_rows, aligned GEMM operands)
        out = (torch.empty if out_cols == 475 else torch.zeros)((B, out_cols), dtype=torch.bfloat16, device=x.device)
        _lib.check(_lib.lib().catan_tile_encoder_fwd_train(_ptr(x), _ptr(wts), _ptr(vecs), _ptr(out), out_cols, C.cast(ptrs, C.c_void_p), B, _stream()))
        ctx.save_for_backward(*saves, *params)
        ctx.eps = float(te.norm.eps)
        ctx.B = B
        return out

    @staticmethod
    def backward(ctx, dout):
        # The saved activations are views of the ONE shared workspace, which autograd's version counters do not track: once this
        # backward has given the lease back, a later forward may overwrite them.  A second backward over the same graph
        # (retain_graph=True) would then silently compute gradients from another step's activations: refuse it.
        if getattr(ctx, "consumed", False) and getattr(ctx, "lease", None) is not None:
            raise RuntimeError("_TileEncoderTrain: second backward over a forward whose activation workspace has been released "
                               "(retain_graph is not supported with the shared workspace; set CATAN_TE_WORKSPACE=0)")
        ctx.consumed = True
        ns = len(ctx.save_names)
        sv = dict(zip(ctx.save_names, ctx.saved_tensors[:ns]))
        P = ctx.saved_tensors[ns:]
        B, eps, bf = ctx.B, ctx.eps, torch.bfloat16
        T = B * 19
        g = [None] * len(P)
        im = _te_images(ctx.te) if weight_images.enabled else None
        L = _lib.lib()
        with torch.autocast("cuda", enabled=False):
            d = _aligned(dout[:, :475].reshape(T, 25).to(bf))
            dp, g[38], g[39] = _ln_backward(sv["p"], P[38], P[39], d, eps, True)
            dx = _rows_product(dp, im.wpt if im is not None else P[36].to(bf).t().contiguous())   # [T, 64]
            g[36], g[37] = _wgrad(sv["xfin"], dp, True)
            for l in (1, 0):
                b = 4 + 16 * l
                xin, n1, qkv, o, xmid, n2, h = (sv.get(k + str(l)) for k in ("xin", "n1_", "qkv", "o", "xmid", "n2_", "h"))
                if im is not None:
                    w2t, w1t, wot, wqt = im.w2t[l], im.w1t[l], im.wot[l], im.wqt[l]
                else:
                    w2t, w1t, wot = P[b + 14].to(bf).t().contiguous(), P[b + 12].to(bf).t().contiguous(), P[b + 8].to(bf).t().contiguous()
                    wqt = torch.cat([P[b + 2], P[b + 4], P[b + 6]], 0).to(bf).t().contiguous()
                if _te_backward_fused_w():
                    # k_ffn_bwd_w: the chain below AND both weight gradients in one pass over the rows (dH never leaves the chip)
                    dxmid = torch.empty_like(xmid)
                    acc = grad_zeros((64 * 128 + 64 + 128 * 64 + 128 + 128 + 64 * 64 + 64,), xmid.device)
                    dw2, db2, dw1, db1, dl = acc[:8192], acc[8192:8256], acc[8256:16448], acc[16448:16576], acc[16576:16704]
                    dwo, dbo = acc[16704:20800], acc[20800:]
                    lw, lb = P[b + 10].detach().float().contiguous(), P[b + 11].detach().float().contiguous()
                    if h is None:                                           # (recompute_h) ... h recomputed from the rows of xmid, with the out-projection's backward
                        do = torch.empty_like(o)
                        wts, vecs = ctx.packed                              # the forward's packed parameters: W1 [128][64] and b1 [128] of layer l
                        w1 = wts[4096 + 32768 * l + 16384:4096 + 32768 * l + 16384 + 8192]
                        b1 = vecs[192 + 704 * l + 512:192 + 704 * l + 640]
                        _lib.check(_lib.lib().catan_ffn_outproj_bwd_rh(_ptr(dx), _ptr(xmid), _ptr(w2t), _ptr(w1t), _ptr(w1), _ptr(b1), _ptr(lw), _ptr(lb), eps, _ptr(dxmid),
                                                                       _ptr(dw2), _ptr(db2), _ptr(dw1), _ptr(db1), _ptr(dl[:64]), _ptr(dl[64:]),
                                                                       _ptr(o), _ptr(wot), _ptr(do), _ptr(dwo), _ptr(dbo), T, _stream()))
                        g[b + 8], g[b + 9] = dwo.view(64, 64), dbo
                    elif os.environ.get("CATAN_TE_BWD_OP", "1") == "1":     # ... and the out-projection's dO and weight gradient from the same rows
                        do = torch.empty_like(o)
                        _lib.check(_lib.lib().catan_ffn_outproj_bwd(_ptr(dx), _ptr(h), _ptr(xmid), _ptr(n2) if n2 is not None else None, _ptr(w2t), _ptr(w1t), _ptr(lw), _ptr(lb), eps, _ptr(dxmid),
                                                                    _ptr(dw2), _ptr(db2), _ptr(dw1), _ptr(db1), _ptr(dl[:64]), _ptr(dl[64:]),
                                                                    _ptr(o), _ptr(wot), _ptr(do), _ptr(dwo), _ptr(dbo), T, _stream()))
                        g[b + 8], g[b + 9] = dwo.view(64, 64), dbo
                    else:
                        do = None
                        _lib.check(_lib.lib().catan_ffn_bwd(_ptr(dx), _ptr(h), _ptr(xmid), _ptr(n2) if n2 is not None else None, _ptr(w2t), _ptr(w1t), _ptr(lw), _ptr(lb), eps, _ptr(dxmid),
                                                            _ptr(dw2), _ptr(db2), _ptr(dw1), _ptr(db1), _ptr(dl[:64]), _ptr(dl[64:]), T, _stream()))
                    g[b + 14], g[b + 15], g[b + 12], g[b + 13], g[b + 10], g[b + 11] = dw2.view(64, 128), db2, dw1.view(128, 64), db1, dl[:64], dl[64:]
                    dh = None
                elif os.environ.get("CATAN_TE_BWD_UNFUSED") == "1":
                    dh = _rows_product(dx, w2t, h, MODE_RELU_MASK)                        # (dx @ w2) where h > 0
                    dn2 = _rows_product(dh, w1t)
                    dxmid, g[b + 10], g[b + 11] = _ln_backward(xmid, P[b + 10], P[b + 11], dn2, eps, False, dres=dx)
                else:                           # the same three steps in one pass over the rows (k_ffn_bwd_dx)
                    dh, dxmid = torch.empty_like(h), torch.empty_like(xmid)
                    dl = grad_zeros((2, 64), h.device)
                    lw = P[b + 10].detach().float().contiguous()                          # (named: alive until the launch is queued)
                    _lib.check(_lib.lib().catan_ffn_bwd_dx(_ptr(dx), _ptr(h), _ptr(xmid), _ptr(w2t), _ptr(w1t), _ptr(lw), eps, _ptr(dh), _ptr(dxmid),
                                                           _ptr(dl[0]), _ptr(dl[1]), T, _stream()))
                    g[b + 10], g[b + 11] = dl[0], dl[1]
                if dh is not None:
                    g[b + 14], g[b + 15] = _wgrad(h, dx, True)
                    g[b + 12], g[b + 13] = _wgrad(n2, dh, True)
                if dh is not None or do is None:
                    do = _rows_product(dxmid, wot)
                    g[b + 8], g[b + 9] = _wgrad(o, dxmid, True)
                dqkv = torch.empty_like(qkv)
                _lib.check(_lib.lib().catan_attention_bwd(_ptr(qkv), None, _ptr(do), _ptr(dqkv), B, 19, 4, 16, 1, _stream()))
                fused_w = _te_backward_fused_w()
                if fused_w:                     # k_qkv_bwd_w: the QKV product's weight gradient and the dX chain in one pass over the rows
                    dx = torch.empty_like(xin)
                    acc = grad_zeros((192 * 64 + 192 + 128,), xin.device)
                    dwq, dbq, dl = acc[:12288].view(192, 64), acc[12288:12480], acc[12480:]
                    lw, lb = P[b].detach().float().contiguous(), P[b + 1].detach().float().contiguous()
                    _lib.check(_lib.lib().catan_qkv_bwd(_ptr(dqkv), _ptr(xin), _ptr(dxmid), _ptr(n1) if n1 is not None else None, _ptr(wqt), _ptr(lw), _ptr(lb), eps,
                                                        _ptr(dx), _ptr(dwq), _ptr(dbq),
                                                        _ptr(dl[:64]), _ptr(dl[64:]), T, _stream()))
                    g[b], g[b + 1] = dl[:64], dl[64:]
                else:
                    dwq, dbq = _wgrad(n1, dqkv, True)
                for k in range(3):
                    g[b + 2 + 2 * k], g[b + 3 + 2 * k] = dwq[64 * k:64 * k + 64], dbq[64 * k:64 * k + 64]
                if fused_w:
                    pass
                elif os.environ.get("CATAN_TE_BWD_UNFUSED") == "1":
                    dn1 = _rows_product(dqkv, wqt)
                    dx, g[b], g[b + 1] = _ln_backward(xin, P[b], P[b + 1], dn1, eps, False, dres=dxmid)
                else:                           # the same two steps in one pass over the rows (k_qkv_bwd_dx)
                    dx = torch.empty_like(xin)
                    dl = grad_zeros((2, 64), xin.device)
                    lw = P[b].detach().float().contiguous()
                    _lib.check(_lib.lib().catan_qkv_bwd_dx(_ptr(dqkv), _ptr(xin), _ptr(dxmid), _ptr(wqt), _ptr(lw), eps, _ptr(dx), _ptr(dl[0]), _ptr(dl[1]), T, _stream()))
                    g[b], g[b + 1] = dl[0], dl[1]
            da0, g[2], g[3] = _ln_backward(sv["a0"], P[2], P[3], dx, eps, True)
            dw0, g[1] = _wgrad(sv["tiles64"], da0, True)
            g[0] = dw0[:, :60]
        if ctx.lease is not None:
            ctx.lease.release()
        return (None, None, None) + tuple(g)


def tile_encoder_train_supported(te, tiles):
    """training on the GPU under bf16 autocast, the reference's sizes; CATAN_TE_TRAIN_UNFUSED=1 keeps the sub-layer kernels"""
    import os
    if not torch.is_grad_enabled() or not tiles.is_cuda or tiles.dim() != 3 or tuple(tiles.shape[1:]) != (19, 60) or tiles.requires_grad:
        return False
    if os.environ.get("CATAN_TE_TRAIN_UNFUSED") == "1":
        return False
    if not (torch.is_autocast_enabled() and torch.get_autocast_dtype("cuda") == torch.bfloat16):
        return False
    return (te.first_layer.weight.shape == (64, 60) and len(te.encoder_layers) == 2 and te.out_proj.weight.shape == (25, 64)
            and te.encoder_layers[0].multi_headed_attention.heads == 4 and te.encoder_layers[0].pointwise_net.linear1.weight.shape == (128, 64)
            and te.first_layer.weight.dtype == torch.float32
            and all(abs(m.eps - 1e-5) < 1e-12 for m in [te.norm, te.norm_2] + [s.norm for l in te.encoder_layers for s in l.sublayers]))


def tile_encoder_train(te, tiles, out_cols=475):
    """tiles [B, 19, 60] -> bf16 [B, out_cols >= 475] (zero beyond column 475) with gradients to the encoder's parameters (see _TileEncoderTrain)"""
    return _TileEncoderTrain.apply(tiles, te, int(out_cols), *_te_params(te))


def head_pack(head, trunk_dim):
    """An action head's parameters in the layout of catan_head_fwd (include/catan_hip.h), cached on the module and re-packed - IN
    PLACE, a captured hipGraph holds the buffers' addresses - when a parameter changed.  trunk_dim: the columns of mlp_1's input
    that the shared trunk product covers; the rest (conditioning columns, head 5's trade features) is W1e."""
    params = [head.mlp_1.weight, head.mlp_2.weight, head.mlp_2.bias, head.norm.weight, head.norm.bias,
              head.distribution.linear.weight, head.distribution.linear.bias]
    stamp = (sum(p._version for p in params), params[0].device, params[0].data_ptr(), trunk_dim)
    cache = getattr(head, "_fused_pack", None)
    if cache is not None and cache[0] == stamp:
        return cache[1], cache[2]
    L = _lib.lib()
    K = head.distribution.linear.weight.shape[0]
    e = head.mlp_1.weight.shape[1] - trunk_dim
    assert head.mlp_2.weight.shape == (128, 128) and K <= 80 and 0 <= e <= 32
    with torch.no_grad():
        w1e = torch.zeros((32, 128), dtype=torch.bfloat16, device=params[0].device)
        if e:
            w1e[:e] = head.mlp_1.weight[:, trunk_dim:].t().to(torch.bfloat16)
        wts = torch.cat([_pad2(head.mlp_2.weight, 128, 128), _pad2(head.distribution.linear.weight, 80, 128), w1e.reshape(-1)]).contiguous()
        rb = lambda b, n: _pad1(b.to(torch.bfloat16), n)             # a Linear bias as bf16 autocast hands it to the GEMM
        vec = torch.cat([_pad1(head.norm.weight, 128), _pad1(head.norm.bias, 128), rb(head.mlp_2.bias, 128),
                         rb(head.distribution.linear.bias, 80)]).contiguous()
    assert wts.numel() == L.catan_head_weight_elems() and vec.numel() == L.catan_head_vec_elems()
    if cache is not None and cache[1].device == wts.device:
        cache[1].copy_(wts); cache[2].copy_(vec)
        wts, vec = cache[1], cache[2]
    head._fused_pack = (stamp, wts, vec)
    return wts, vec


def head_fused_supported(pre_all):
    """inference (no autograd) on the GPU with the trunk product in bf16"""
    return (not torch.is_grad_enabled()) and pre_all.is_cuda and pre_all.dtype == torch.bfloat16 and pre_all.stride(-1) == 1 \
        and pre_all.stride(0) % 8 == 0 and fused_heads_enabled


fused_heads_enabled = True
chained_heads_enabled = True       # the heads' glue inside the fused kernels (heads_chain); off: one kernel per head + torch glue


def head_sample(head, trunk_dim, pre, cond, mask, deterministic=False, generator=None):
    """One head evaluation as one kernel (catan_head_fwd).  pre: bf16 [B, 128] view of the shared trunk product (row pitch =
    stride(0)); cond: [B, e] conditioning columns or None; mask float [B, K] (a column window is fine).
    -> (action int64 [B], log-prob [B])."""
    wts, vec = head_pack(head, trunk_dim)
    B = pre.shape[0]
    K = head.distribution.linear.weight.shape[0]
    if mask.stride(-1) != 1 or mask.dtype != torch.float32:
        mask = mask.float().contiguous()
    u = None
    if not deterministic:
        u = generator.take(B) if isinstance(generator, UniformPool) else torch.rand(B, device=pre.device, generator=generator)
    ncond = 0
    if cond is not None:
        cond = cond.float()
        if cond.stride(-1) != 1:
            cond = cond.contiguous()
        ncond = cond.shape[1]
    assert ncond == head.mlp_1.weight.shape[1] - trunk_dim and mask.shape[1] == K
    action = torch.empty(B, dtype=torch.int64, device=pre.device)
    logp = torch.empty(B, dtype=torch.float32, device=pre.device)
    _lib.check(_lib.lib().catan_head_fwd(_ptr(pre), pre.stride(0), _ptr(cond), cond.stride(0) if cond is not None else 0, ncond, _ptr(wts), _ptr(vec),
                                         float(head.norm.eps), K, _ptr(mask), mask.stride(0), _ptr(u), _ptr(action), _ptr(logp), B, _stream()))
    return action, logp


HEAD_CHAIN_ORDER = ((0, 0), (1, 0), (2, 0), (3, 0), (5, 0), (6, 0), (11, 0), (4, 0), (9, 0), (10, 0),
                    (7, 0), (7, 1), (7, 2), (7, 3), (8, 0), (8, 1), (8, 2), (8, 3))       # (head, step): the order of the pass's 18 draws


def head5_custom_pack(head):
    """head 5's custom_mlp + custom_norm as catan_head_chain reads them (fp32 [480]; W and b rounded to bf16 as autocast hands
    them to the GEMM), cached and refreshed in place like head_pack"""
    params = [head.custom_mlp.weight, head.custom_mlp.bias, head.custom_norm.weight, head.custom_norm.bias]
    stamp = (sum(p._version for p in params), params[0].device, params[0].data_ptr())
    cache = getattr(head, "_custom_pack", None)
    if cache is not None and cache[0] == stamp:
        return cache[1]
    with torch.no_grad():
        r = lambda t: t.to(torch.bfloat16).float().reshape(-1)
        pack = torch.cat([r(head.custom_mlp.weight), r(head.custom_mlp.bias), head.custom_norm.weight.float(), head.custom_norm.bias.float()]).contiguous()
    assert pack.numel() == 480 and float(head.custom_norm.eps) == float(head.norm.eps)
    if cache is not None and cache[1].device == pack.device:
        cache[1].copy_(pack)
        pack = cache[1]
    head._custom_pack = (stamp, pack)
    return pack


def heads_chain(heads, trunk_dim, pre_all, masks, cur_res, trade, deterministic=False, generator=None, forced_type=None):
    """All twelve heads of an inference pass - eighteen head evaluations - as eighteen launches of the fused head kernel in its
    chained mode (catan_head_chain): the glue between them (type-conditional mask rows, conditioning columns, log-prob masks,
    the trade heads' lists) runs inside the kernels on a per-row state.  -> (actions int64 [B, 18], joint log-prob [B])"""
    L = _lib.lib()
    B, dev = pre_all.shape[0], pre_all.device
    masks = masks if (masks.dtype == torch.float32 and masks.is_contiguous()) else masks.float().contiguous()
    cur_res = cur_res.float().contiguous(); trade = trade.float().contiguous()
    state = torch.zeros((B, L.catan_head_state_floats()), dtype=torch.float32, device=dev)
    actions = torch.empty((B, 18), dtype=torch.int64, device=dev)
    logp = torch.empty((B,), dtype=torch.float32, device=dev)
    forced = None if forced_type is None else forced_type.to(torch.int64).contiguous()
    us = None
    if not deterministic:
        if isinstance(generator, UniformPool):
            us = [generator.take(B) for _ in HEAD_CHAIN_ORDER]
        else:
            u_all = torch.rand((len(HEAD_CHAIN_ORDER), B), device=dev, generator=generator)
            us = [u_all[k] for k in range(len(HEAD_CHAIN_ORDER))]
    custom = head5_custom_pack(heads[5])
    st = _stream()
    for k, (h, step) in enumerate(HEAD_CHAIN_ORDER):
        wts, vec = head_pack(heads[h], trunk_dim)
        pre = pre_all[:, 128 * h:128 * (h + 1)]
        _lib.check(L.catan_head_chain(_ptr(pre), pre_all.stride(0), _ptr(wts), _ptr(vec), float(heads[h].norm.eps), h, step, _ptr(state), _ptr(masks),
                                      _ptr(cur_res), _ptr(trade), _ptr(custom) if h == 5 else None, _ptr(forced) if h == 0 else None,
                                      None if us is None else _ptr(us[k]), _ptr(actions), _ptr(logp), B, st))
    return actions, logp


_PATTERN_LISTS = {}


def _pattern_lists(device):
    """the synthetic list of every count pattern (include/catan_hip.h): ids int8 [P, 25] and lens int32 [P]"""
    if device not in _PATTERN_LISTS:
        P = _lib.lib().catan_card_summary_patterns()
        k = torch.arange(P)
        counts = torch.stack((k % 2, k // 2 % 15, k // 30 % 6, k // 180 % 3, k // 540 % 3, k // 1620), 1)      # [P, 6]
        lens = counts.sum(1).clamp(max=25)                       # (patterns with more than 25 cards cannot occur; their dout is zero)
        ids = torch.repeat_interleave(torch.arange(6).repeat(P), counts.reshape(-1)).split(counts.sum(1).tolist())
        rows = torch.zeros((P, 25), dtype=torch.int8)
        for i, r in enumerate(ids):
            rows[i, :min(25, r.numel())] = r[:25].to(torch.int8)
        _PATTERN_LISTS[device] = (rows.to(device), lens.to(device=device, dtype=torch.int32))
    return _PATTERN_LISTS[device]


class _CardSummary(torch.autograd.Function):
    """k_card_summary_fwd / _bwd (csrc/catan_nn.hip): the dev-card list module evaluated per card class from small tables.
    Backward: the gradient of the tables is linear in dout and a list's Jacobian depends only on its count PATTERN (4 860
    possible ones), so dout is summed per pattern and the 4 860 synthetic lists are differentiated; lists whose counts fall
    outside the deck (key -1) go through the kernel directly."""

    @staticmethod
    def forward(ctx, ids, lens, params, eps):
        rows = ids.shape[0]
        out = torch.empty((rows, 16), dtype=torch.float32, device=ids.device)
        keys = torch.empty((rows,), dtype=torch.int32, device=ids.device)
        p = params.detach().contiguous()
        _lib.check(_lib.lib().catan_card_summary_fwd(_ptr(ids), ids.element_size(), ids.stride(0), _ptr(lens), _ptr(p), float(eps), _ptr(out), _ptr(keys),
                                                     rows, _stream()))
        ctx.save_for_backward(ids, lens, p, keys)
        ctx.eps = float(eps)
        return out

    @staticmethod
    def backward(ctx, dout):
        ids, lens, p, keys = ctx.saved_tensors
        L = _lib.lib()
        dparams = grad_zeros(tuple(p.shape), p.device) if p.dtype == torch.float32 else torch.zeros_like(p)
        d = dout.contiguous().float()
        pid, plen = _pattern_lists(ids.device)
        reps = 8 if ids.shape[0] >= 32768 else 1
        dpat = grad_zeros((reps, pid.shape[0], 16), ids.device)
        n_unkeyed = torch.zeros((1,), dtype=torch.int32, device=ids.device)
        _lib.check(L.catan_card_pattern_sum(_ptr(keys), _ptr(d), _ptr(dpat), reps, _ptr(n_unkeyed), ids.shape[0], _stream()))
        dpat = dpat.sum(0) if reps > 1 else dpat[0]
        _lib.check(L.catan_card_summary_bwd(_ptr(pid), 1, pid.stride(0), _ptr(plen), _ptr(p), ctx.eps, _ptr(dpat), _ptr(dparams), None, None, pid.shape[0], _stream()))
        _lib.check(L.catan_card_summary_bwd(_ptr(ids), ids.element_size(), ids.stride(0), _ptr(lens), _ptr(p), ctx.eps, _ptr(d), _ptr(dparams), _ptr(keys),
                                            _ptr(n_unkeyed), ids.shape[0], _stream()))
        return None, None, dparams, None


def card_summary_supported(ids, vocab, heads, hd, width):
    return ids.is_cuda and ids.dim() == 2 and ids.shape[1] <= 25 and (vocab, heads, hd, width) == (6, 4, 4, 16) \
        and ids.dtype in (torch.int8, torch.int32, torch.int64)


def card_summary(ids, lens, params, eps):
    """ids [rows, L<=25] integer (any row pitch), lens [rows], params float32 [544] (S, V, W, bias, LayerNorm weight / bias:
    include/catan_hip.h) -> float32 [rows, 16]; gradient w.r.t. params."""
    if ids.stride(1) != 1:
        ids = ids.contiguous()
    return _CardSummary.apply(ids, lens.to(torch.int32).contiguous(), params, eps)


import weakref
_CARD_TABLES = weakref.WeakKeyDictionary()


def card_summary_params(embedding, mha, norm):
    """the 544 floats catan_card_summary_* read (include/catan_hip.h), from the module's weights"""
    import math
    V, H, hd = embedding.num_embeddings, mha.heads, mha.hd
    w = torch.cat([n.weight for n in mha.qkv_nets], 0).float()
    b = torch.cat([n.bias for n in mha.qkv_nets], 0).float()
    qkv = torch.nn.functional.linear(embedding.weight.float(), w, b).view(V, 3, H, hd)      # Q / K / V of each of the six ids
    s_ab = torch.einsum("ahd,bhd->hab", qkv[:, 0], qkv[:, 1]) * (1.0 / math.sqrt(hd))
    return torch.cat((s_ab.reshape(-1), qkv[:, 2].reshape(-1), mha.out_proj_net.weight.float().reshape(-1),
                      mha.out_proj_net.bias.float(), norm.weight.float(), norm.bias.float()))


def card_summary_table(embedding, mha, norm):
    """(params [544], table [patterns, 16]) for inference, cached on the attention module per LayerNorm and rebuilt - IN PLACE: a
    captured hipGraph holds the addresses - when a weight changed (the inference copy is refreshed once per rollout)."""
    prm = [embedding.weight, norm.weight, norm.bias] + list(mha.parameters())
    stamp = (sum(p._version for p in prm), prm[0].device, prm[0].data_ptr())
    caches = _CARD_TABLES.setdefault(mha, {})              # (not on the module: a deepcopy - inference_copy - must not carry them along)
    cache = caches.get(id(norm))
    if cache is not None and cache[0] == stamp:
        return cache[1], cache[2]
    with torch.no_grad(), torch.autocast(device_type="cuda", enabled=False):
        params = card_summary_params(embedding, mha, norm).contiguous()
        pid, plen = _pattern_lists(params.device)
        table = torch.empty((pid.shape[0], 16), dtype=torch.float32, device=params.device)
        _lib.check(_lib.lib().catan_card_summary_fwd(_ptr(pid), 1, pid.stride(0), _ptr(plen), _ptr(params), float(norm.eps), _ptr(table), None,
                                                     pid.shape[0], _stream()))
    if cache is not None and cache[1].device == params.device:
        cache[1].copy_(params); cache[2].copy_(table)
        params, table = cache[1], cache[2]
    caches[id(norm)] = (stamp, params, table, embedding, norm)
    return params, table


def card_summary_lookup(ids, lens, embedding, mha, norm):
    """inference form of card_summary: pattern table look-up (k_card_summary_lookup)"""
    params, table = card_summary_table(embedding, mha, norm)
    if ids.stride(1) != 1:
        ids = ids.contiguous()
    lens = lens.to(torch.int32).contiguous()
    out = torch.empty((ids.shape[0], 16), dtype=torch.float32, device=ids.device)
    _lib.check(_lib.lib().catan_card_summary_lookup(_ptr(ids), ids.element_size(), ids.stride(0), _ptr(lens), _ptr(table), _ptr(params), float(norm.eps),
                                                    _ptr(out), ids.shape[0], _stream()))
    return out


def refresh_card_tables(module):
    """brings every cached pattern table under `module` up to date in place (for captured graphs: policy.refresh_kernel_packs)"""
    for m in module.modules():
        for cache in list(_CARD_TABLES.get(m, {}).values()):
            card_summary_table(cache[3], m, cache[4])


class _MaskedCategorical(torch.autograd.Function):
    """csrc/catan_nn.hip k_categorical_fwd / _bwd: one launch for log_softmax(logits + log(mask)), the action (given /
    arg-max / inverse-CDF sample), its log-prob and the entropy."""

    @staticmethod
    def forward(ctx, logits, mask, given, u):
        B, K = logits.shape
        logits = logits.contiguous()
        if mask.stride(-1) != 1 or mask.dtype != torch.float32:
            mask = mask.float().contiguous()
        action = torch.empty(B, dtype=torch.int64, device=logits.device)
        logp = torch.empty(B, dtype=torch.float32, device=logits.device)
        ent, lse = torch.empty_like(logp), torch.empty_like(logp)
        _lib.check(_lib.lib().catan_categorical_fwd(_ptr(logits), _ptr(mask), mask.stride(0), _ptr(given), _ptr(u), _ptr(action), _ptr(logp),
                                                    _ptr(ent), _ptr(lse), B, K, _stream()))
        ctx.save_for_backward(logits, mask, action, lse, ent)
        ctx.mark_non_differentiable(action)
        return action, logp, ent

    @staticmethod
    def backward(ctx, _da, dlogp, dent):
        logits, mask, action, lse, ent = ctx.saved_tensors
        B, K = logits.shape
        dlogits = torch.empty_like(logits)
        dl, de = dlogp.float().contiguous(), dent.float().contiguous()        # (named: two temporaries would share one freed block)
        _lib.check(_lib.lib().catan_categorical_bwd(_ptr(logits), _ptr(mask), mask.stride(0), _ptr(action), _ptr(lse), _ptr(ent),
                                                    _ptr(dl), _ptr(de), _ptr(dlogits), B, K, _stream()))
        return dlogits, None, None, None


# the learner's heads read packed mask bits (policy._evaluate_compact); "0": the float mask rows expanded and gathered per minibatch step
CATEGORICAL_BITS = os.environ.get("CATAN_CATEGORICAL_BITS", "1") != "0"


class _MaskedCategoricalBits(torch.autograd.Function):
    """k_categorical_bits_fwd / _bwd: _MaskedCategorical for given actions with the mask as bits of the packed mask rows (no float mask matrix)"""

    @staticmethod
    def forward(ctx, logits, packed, rows_idx, segs, given, given_ld):
        import ctypes as C
        B, K = logits.shape
        logits = logits.contiguous()
        action = torch.empty(B, dtype=torch.int64, device=logits.device)
        logp = torch.empty(B, dtype=torch.float32, device=logits.device)
        ent, lse = torch.empty_like(logp), torch.empty_like(logp)
        sg = (C.c_int32 * 8)(*segs)
        _lib.check(_lib.lib().catan_categorical_bits_fwd(_ptr(logits), _ptr(packed), packed.stride(0), _ptr(rows_idx) if rows_idx is not None else None, C.cast(sg, C.c_void_p),
                                                         C.c_void_p(given.data_ptr()), given_ld, _ptr(action), _ptr(logp), _ptr(ent), _ptr(lse), B, K, _stream()))
        ctx.save_for_backward(logits, packed, action, lse, ent, *(() if rows_idx is None else (rows_idx,)))
        ctx.segs = tuple(segs)
        ctx.mark_non_differentiable(action)
        return action, logp, ent

    @staticmethod
    def backward(ctx, _da, dlogp, dent):
        import ctypes as C
        logits, packed, action, lse, ent = ctx.saved_tensors[:5]
        rows_idx = ctx.saved_tensors[5] if len(ctx.saved_tensors) > 5 else None
        B, K = logits.shape
        dlogits = torch.empty_like(logits)
        dl, de = dlogp.float().contiguous(), dent.float().contiguous()
        sg = (C.c_int32 * 8)(*ctx.segs)
        _lib.check(_lib.lib().catan_categorical_bits_bwd(_ptr(logits), _ptr(packed), packed.stride(0), _ptr(rows_idx) if rows_idx is not None else None, C.cast(sg, C.c_void_p),
                                                         _ptr(action), _ptr(lse), _ptr(ent), _ptr(dl), _ptr(de), _ptr(dlogits), B, K, _stream()))
        return dlogits, None, None, None, None, None


def masked_categorical_bits(logits, packed, rows_idx, segments, given):
    """logits fp32 [B, K]; packed int32 [n, pitch] (the env's mask rows); rows_idx int64 [B] (row j -> packed row) or None; segments: a list
    of 1..3 (row count, bit offset, AND offset or None) - consecutive row ranges of the launch; given: int64 [B], any stride (a column of an
    action matrix).  -> (action, log-prob, entropy) as masked_categorical."""
    B = logits.shape[0]
    segs = list(segments) + [(0, segments[-1][1], segments[-1][2])] * (3 - len(segments))
    n0 = segs[0][0]
    n1 = n0 + segs[1][0] if len(segments) > 1 else B
    if len(segments) == 1:
        n0 = B
    flat = [n0, n1]
    for _, off, aoff in segs:
        flat += [int(off), -1 if aoff is None else int(aoff)]
    assert given.dtype == torch.int64 and given.dim() == 1 and packed.dtype == torch.int32 and packed.stride(1) == 1
    return _MaskedCategoricalBits.apply(logits, packed, rows_idx, flat, given, given.stride(0) if B > 1 else 1)


def categorical_supported(logits):
    return logits.is_cuda and logits.dtype == torch.float32 and logits.dim() == 2


def recurrent_given(acts, cur_res, fixed, from_hand, dtype):
    """catan_recurrent_given: acts int64 [B, >= 4] (any row stride), cur_res float [B, 6], fixed float [B, kf] or None ->
    (cond [4B, kf + 6] in `dtype`, mask [4B, 6], given int64 [4B], keep [B, 4], out_final [B, 6])"""
    B, dev = acts.shape[0], acts.device
    if acts.stride(1) != 1:
        acts = acts.contiguous()
    cr = cur_res.float().contiguous()
    fx = None if fixed is None else fixed.float().contiguous()
    kf = 0 if fx is None else fx.shape[1]
    bf = dtype == torch.bfloat16
    cond = torch.empty((4 * B, kf + 6), dtype=torch.bfloat16 if bf else torch.float32, device=dev)
    mask = torch.empty((4 * B, 6), dtype=torch.float32, device=dev)
    given = torch.empty((4 * B,), dtype=torch.int64, device=dev)
    keep = torch.empty((B, 4), dtype=torch.float32, device=dev)
    outf = torch.empty((B, 6), dtype=torch.float32, device=dev)
    if B:
        _lib.check(_lib.lib().catan_recurrent_given(_ptr(acts), acts.stride(0), _ptr(cr), _ptr(fx) if fx is not None else None, kf, int(bool(from_hand)), B, int(bf),
                                                    _ptr(cond), _ptr(mask), _ptr(given), _ptr(keep), _ptr(outf), _stream()))
    return cond, mask, given, keep, outf


class UniformPool(object):
    """The uniforms of all the categorical draws of one policy pass from ONE `torch.rand` (a pass makes 18 draws; at rollout
    width every launch counts): pass it where a generator is expected; rows are handed out in call order."""

    def __init__(self, generator, rows, draws, device):
        self.u = torch.rand((draws, rows), device=device, generator=generator)
        self.generator, self.k = generator, 0

    def take(self, rows):
        if self.k < self.u.shape[0] and rows == self.u.shape[1]:
            self.k += 1
            return self.u[self.k - 1]
        return torch.rand(rows, device=self.u.device, generator=self.generator)


def masked_categorical(logits, mask, given=None, deterministic=False, generator=None):
    """logits fp32 [B,K]; mask [B,K] (a column window of the mask matrix is fine); given int64 [B] or None.
    -> (action int64 [B], log-prob of the action [B], entropy [B])."""
    B = logits.shape[0]
    u = None
    if given is None and not deterministic:
        u = generator.take(B) if isinstance(generator, UniformPool) else torch.rand(B, device=logits.device, generator=generator)
    if given is not None:
        given = given.contiguous()
    return _MaskedCategorical.apply(logits, mask, given, u)
