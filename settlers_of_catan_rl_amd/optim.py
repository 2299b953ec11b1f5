"""The optimiser step of `PPO.update` - `nn.utils.clip_grad_norm_(parameters, max_grad_norm)` followed by `optim.Adam(lr, eps).step()`
(reference RL/ppo/ppo.py:23,67-68) - as ONE call over all parameters.

On the device it is two launches of `catan_adam_step` (csrc/catan_optim.hip: per-chunk sums of squared gradients; norm, clip
coefficient and the Adam update per chunk) instead of clip_grad_norm_'s three and Adam's nine `_foreach_*` launches over ~300 tensors
(0.55 ms -> ~0.05 ms of a minibatch step).  The exponential averages live in two flat buffers, the gradients stay wherever autograd
(or the flat all-reduce bucket, dist.GradBucket) put them: only their addresses are handed over, through a pinned host array, each
step.  On the CPU the same update runs as torch operations (what the tests compare with torch.optim.Adam)."""
import ctypes as C
import math

import torch


class FusedAdam(object):
    """`torch.optim.Adam(params, lr, betas, eps)` (no weight decay, no amsgrad) + gradient-norm clipping in `step(max_grad_norm)`.
    `param_groups[0]["lr"]` is read at every step (train_loop's linear decay writes it, as RL/robust_train.py:67-72 does)."""
    RING = 8

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8):
        self.params = [p for p in params if p.requires_grad]
        if not self.params:
            raise ValueError("FusedAdam needs at least one parameter")
        dev = self.params[0].device
        if any(p.device != dev or p.dtype != torch.float32 or not p.is_contiguous() for p in self.params):
            raise ValueError("FusedAdam: contiguous fp32 parameters on one device")
        self.param_groups = [{"params": self.params, "lr": float(lr), "betas": (float(betas[0]), float(betas[1])), "eps": float(eps)}]
        self.device, self.steps = dev, 0
        # every tensor's slice of the flat state starts on a 16-byte boundary
        offs, total = [], 0
        for p in self.params:
            offs.append(total)
            total += (p.numel() + 3) & ~3
        self.exp_avg = torch.zeros(total, dtype=torch.float32, device=dev)
        self.exp_avg_sq = torch.zeros(total, dtype=torch.float32, device=dev)
        self._m = [self.exp_avg[o:o + p.numel()].view_as(p) for o, p in zip(offs, self.params)]
        self._v = [self.exp_avg_sq[o:o + p.numel()].view_as(p) for o, p in zip(offs, self.params)]
        self.last_norm = None
        self._tables = None
        self.copied_grads = 0            # gradients that went through a temporary aligned fp32 copy (expected 0: tests / diagnostics)

    # ---- the device tables (built once; rebuilt if a parameter's storage moved)
    def _build_tables(self):
        import numpy as np
        from . import _lib
        L = _lib.lib()
        CH = int(L.catan_adam_chunk_elements())
        tens = np.zeros((len(self.params), 3), dtype=np.uint64)
        chunks = []
        for i, (p, m, v) in enumerate(zip(self.params, self._m, self._v)):
            if p.data_ptr() % 16:
                raise ValueError("FusedAdam: a parameter is not 16-byte aligned")
            tens[i] = (p.data_ptr(), m.data_ptr(), v.data_ptr())
            for o in range(0, p.numel(), CH):
                chunks.append((i, min(CH, p.numel() - o), o))
        ch = np.zeros(len(chunks), dtype=np.dtype([("tensor", np.int32), ("count", np.int32), ("offset", np.int64)]))
        for k, c in enumerate(chunks):
            ch[k] = c
        dev = self.device
        self._tables = {
            "sig": tuple(p.data_ptr() for p in self.params),
            "tensors": torch.from_numpy(tens.view(np.uint8).reshape(-1)).to(dev),
            "chunks": torch.from_numpy(ch.view(np.uint8).reshape(-1).copy()).to(dev), "n_chunks": len(chunks),
            # this step's gradient addresses travel through a RING of pinned host rows (the host runs several steps ahead of the device: one
            # row would be rewritten for step k + 1 before the copy engine has read step k's - the kernel would then take step k + 1's
            # gradient buffers, not yet written: round 5, found as NaN losses in the third update of bench.py) - a row is reused only
            # after the event recorded behind its copy has completed
            "grads_host": torch.zeros((self.RING, len(self.params)), dtype=torch.int64).pin_memory(),
            "grads": torch.zeros((self.RING, len(self.params)), dtype=torch.int64, device=dev),
            "grads_host_np": None,
            "events": [None] * self.RING, "slot": 0,
            "partial": torch.zeros(len(chunks), dtype=torch.float64, device=dev),
            "norm": torch.zeros(1, dtype=torch.float32, device=dev),
        }

    def zero_grad(self, set_to_none=True):
        for p in self.params:
            if set_to_none:
                p.grad = None
            elif p.grad is not None:
                p.grad.zero_()

    @torch.no_grad()
    def step(self, max_grad_norm=None):
        """clip_grad_norm_(max_grad_norm) (None / <= 0: no clipping) + Adam.  The gradient tensors are read, not modified (torch's
        clip scales them in place; nothing reads them after the step).  -> None; `last_norm` holds the norm (a device scalar)."""
        g = self.param_groups[0]
        lr, (b1, b2), eps = float(g["lr"]), g["betas"], float(g["eps"])
        self.steps += 1
        bc1 = 1.0 - b1 ** self.steps
        bc2_sqrt = math.sqrt(1.0 - b2 ** self.steps)
        clip = float(max_grad_norm) if (max_grad_norm is not None and max_grad_norm > 0) else 0.0
        if self.device.type != "cuda":
            return self._step_torch(lr, b1, b2, eps, bc1, bc2_sqrt, clip)
        from . import _lib
        if self._tables is None or self._tables["sig"] != tuple(p.data_ptr() for p in self.params):
            self._build_tables()
        t = self._tables
        if t["grads_host_np"] is None:
            t["grads_host_np"] = t["grads_host"].numpy()      # shares the pinned memory
        k = t["slot"] = (t["slot"] + 1) % self.RING
        if t["events"][k] is not None:
            t["events"][k].synchronize()                      # the copy (and the step) that last used this row is done
        ptrs, keep = [], []
        for p in self.params:
            gr = p.grad
            if gr is None:
                ptrs.append(0)
                continue
            if gr.dtype != torch.float32 or not gr.is_contiguous() or gr.data_ptr() % 16:
                # the kernels read 16-byte vectors of fp32: anything else goes through a temporary copy.  p.grad itself is never replaced
                # (it may be a view of dist.GradBucket's flat buffer - whose views are 16-byte aligned, so that path does not come here)
                gr = gr.float().contiguous().clone()
                keep.append(gr)          # (freed after the launch is queued: the caching allocator keeps it valid for this stream's kernels)
                self.copied_grads += 1
            ptrs.append(gr.data_ptr())
        t["grads_host_np"][k, :] = ptrs                       # (one numpy assignment: element-wise writes into the tensor were ~300 aten ops per step)
        host = t["grads_host"][k]
        t["grads"][k].copy_(host, non_blocking=True)
        P = lambda x: C.c_void_p(x.data_ptr())
        _lib.check(_lib.lib().catan_adam_step(P(t["tensors"]), P(t["chunks"]), t["n_chunks"], P(t["grads"][k]), P(t["partial"]), clip, lr, b1, b2, eps,
                                              bc1, bc2_sqrt, P(t["norm"]), C.c_void_p(torch.cuda.current_stream().cuda_stream)))
        if t["events"][k] is None:
            t["events"][k] = torch.cuda.Event()
        t["events"][k].record()
        self.last_norm = t["norm"]
        # the kernel wrote the parameters behind autograd's back: bump their version counters as an in-place torch update would (caches of
        # derived data are keyed on them: nn_kernels.tile_encoder_pack, weight_images, an inference copy's refresh)
        torch.autograd.graph.increment_version([p for p in self.params if p.grad is not None])

    def _step_torch(self, lr, b1, b2, eps, bc1, bc2_sqrt, clip):
        grads = [(p, m, v, p.grad) for p, m, v in zip(self.params, self._m, self._v) if p.grad is not None]
        if not grads:
            return
        total = torch.sqrt(sum((gr.double() ** 2).sum() for _, _, _, gr in grads)).float()
        coef = torch.clamp(clip / (total + 1e-6), max=1.0) if clip > 0 else torch.ones(())
        for p, m, v, gr in grads:
            gc = gr * coef
            m.lerp_(gc, 1.0 - b1)
            v.mul_(b2).addcmul_(gc, gc, value=1.0 - b2)
            p.addcdiv_(m, (v.sqrt() / bc2_sqrt).add_(eps), value=-(lr / bc1))
        self.last_norm = total.reshape(1)

    # ---- checkpoints (torch.optim.Adam's layout: state per parameter index)
    def state_dict(self):
        return {"state": {i: {"step": torch.tensor(float(self.steps)), "exp_avg": m.detach().clone(), "exp_avg_sq": v.detach().clone()}
                          for i, (m, v) in enumerate(zip(self._m, self._v))},
                "param_groups": [{k: v for k, v in self.param_groups[0].items() if k != "params"} | {"params": list(range(len(self.params)))}]}

    def load_state_dict(self, sd):
        # one step counter for all parameters (torch.optim.Adam keeps one per parameter; they only differ there when a parameter had no
        # gradient in some step - under a multi-rank GradBucket every parameter has one in every step): refuse a checkpoint whose counters
        # disagree or whose parameter list is not this optimiser's instead of loading something subtly different
        if len(sd["state"]) not in (0, len(self.params)):
            raise ValueError(f"FusedAdam.load_state_dict: {len(sd['state'])} parameter states for {len(self.params)} parameters")
        steps = {int(float(s["step"])) for s in sd["state"].values()}
        if len(steps) > 1:
            raise ValueError(f"FusedAdam.load_state_dict: per-parameter step counts differ ({sorted(steps)[:4]} ...): not a state this optimiser can continue")
        for i, s in sd["state"].items():
            if tuple(s["exp_avg"].shape) != tuple(self._m[int(i)].shape):
                raise ValueError(f"FusedAdam.load_state_dict: state {i} has shape {tuple(s['exp_avg'].shape)}, the parameter {tuple(self._m[int(i)].shape)}")
            self._m[int(i)].copy_(s["exp_avg"]); self._v[int(i)].copy_(s["exp_avg_sq"])
            self.steps = int(float(s["step"]))
        for k in ("lr", "betas", "eps"):
            if k in sd["param_groups"][0]:
                self.param_groups[0][k] = sd["param_groups"][0][k]
