"""Flat parameter / gradient buffers and the fused Adam step over them.

Every network's parameters are views into ONE contiguous fp32 buffer and their .grad
fields are views into a second one, so the optimiser is a single kernel per network and
data-parallel training all-reduces one bucket per network.
Semantics of torch.optim.Adam as the reference uses it
(/root/reference/models/sinskitG_model.py:589-599: betas=(0.0, 0.99), eps 1e-8, no weight decay).
The step counter and the learning rate live in device memory so that a captured HIP graph
of the whole training step stays valid from one iteration (and one LR-schedule epoch) to the next.
"""
import torch

from . import ops


class FlatParams:
    def __init__(self, module, first=None):
        """first: optional predicate on parameter names; matching parameters are laid out first in the flat buffers and
        `self.split` is the offset where the rest starts (two contiguous gradient buckets: the generator's decoder gradients are
        complete halfway through its backward and can be all-reduced under the encoder's).  state_dict order is unaffected."""
        named = list(module.named_parameters())
        if first is not None:
            named = [kv for kv in named if first(kv[0])] + [kv for kv in named if not first(kv[0])]
        self.split = sum(p.numel() for k, p in named if first(k)) if first is not None else None
        params = [p for _, p in named]
        dev = params[0].device
        total = sum(p.numel() for p in params)
        self.flat = torch.empty(total, dtype=torch.float32, device=dev)
        self.grad = torch.zeros(total, dtype=torch.float32, device=dev)
        self.params = params
        o = 0
        for p in params:
            n = p.numel()
            view = self.flat[o:o + n].view_as(p)
            view.copy_(p.data)
            p.data = view
            p.grad = self.grad[o:o + n].view_as(p)
            p.requires_grad_(False)  # no autograd anywhere on the hot path
            o += n
        self.numel = total

    def buckets(self, name):
        """gradient buckets of this network for the data-parallel all-reduce: {bucket name: contiguous view of the flat gradient}"""
        if getattr(self, "cuts", None):
            edges = [0] + list(self.cuts) + [self.numel]
            return {"%s_%d" % (name, k): self.grad[edges[k]:edges[k + 1]] for k in range(len(edges) - 1)}
        if self.split is None or self.split in (0, self.numel):
            return {name: self.grad}
        return {name + "_dec": self.grad[:self.split], name + "_enc": self.grad[self.split:]}

    def chunk(self, k, min_floats=1 << 22):
        """Cut the flat gradient into <= k contiguous buckets of roughly equal size whose boundaries are STARTS OF CONVOLUTION WEIGHTS
        (4-d parameters): a backward pass completes the gradients from the end of the buffer towards its start, layer by layer, so
        bucket j is complete as soon as the backward has passed the layer its first weight belongs to, and its all-reduce can travel
        under the rest of the backward (the pix2pixHD generator's 730 MB gradient in >= 8 pieces).  Buckets smaller than
        min_floats are not worth a collective of their own.  Returns the cut parameters (the tensors whose start opens buckets 1 ..)."""
        starts, o = [], 0
        for p in self.params:
            if p.dim() == 4 and o > 0:
                starts.append((o, p))
            o += p.numel()
        k = max(1, min(int(k), self.numel // max(1, int(min_floats))))
        cuts, cut_params = [], []
        for j in range(1, k):
            target = self.numel * j // k
            cand = min(starts, key=lambda s: abs(s[0] - target)) if starts else None
            if cand is not None and cand[0] not in cuts and (not cuts or cand[0] > cuts[-1]):
                cuts.append(cand[0])
                cut_params.append(cand[1])
        self.cuts, self.cut_params = cuts, cut_params
        return cut_params


class FlatAdam(torch.optim.Optimizer):
    """Adam over a FlatParams.  A torch Optimizer subclass only so that the reference's LR
    schedulers (networks.get_scheduler) can drive `param_groups[0]["lr"]`; the update itself
    is one vts_adam_flat_dev launch."""

    def __init__(self, flat, lr, betas=(0.9, 0.999), eps=1e-8, step_dev=None, span=None):
        """step_dev: optional 1-element int32 device view that holds this optimiser's step counter (a model may keep the counters of
        all its optimisers in one tensor and advance them with ONE launch per step: step(bump=False)).
        span: (lo, hi) element range of the flat buffer this optimiser updates -- an optimiser built over a subset of the parameters
        (pix2pixHD --niter_fix_global: only the last local enhancer, pix2pixHD_model.py:403-421); the rest is never touched."""
        super().__init__(flat.params, dict(lr=lr, betas=betas, eps=eps))
        self.flat = flat
        self.span = (0, flat.numel) if span is None else (int(span[0]), int(span[1]))
        assert 0 <= self.span[0] < self.span[1] <= flat.numel
        dev = flat.flat.device
        self.m = torch.zeros_like(flat.flat)
        self.v = torch.zeros_like(flat.flat)
        self.step_count = 0
        self.step_dev = step_dev if step_dev is not None else torch.zeros(1, dtype=torch.int32, device=dev)
        self.lr_dev = torch.full((1,), float(lr), dtype=torch.float32, device=dev)
        self._lr_on_dev = float(lr)

    def zero_grad(self, set_to_none=True):
        # gradients are overwritten (not accumulated) by the first backward of every step
        pass

    def sync_lr(self):
        """Push a scheduler-updated learning rate to the device scalar (call outside graph capture)."""
        lr = float(self.param_groups[0]["lr"])
        if lr != self._lr_on_dev:
            self.lr_dev.fill_(lr)
            self._lr_on_dev = lr

    @torch.no_grad()
    def step(self, grad_scale=1.0, closure=None, bump=True):
        """Capturable: increments the device step counter (bump=False: the owner of the counter already did) and launches the fused update."""
        self.step_count += 1
        if bump:
            self.step_dev.add_(1)
        g = self.param_groups[0]
        lo, hi = self.span
        ops.adam_flat_dev(self.flat.flat[lo:hi], self.flat.grad[lo:hi], self.m[lo:hi], self.v[lo:hi], self.lr_dev, g["betas"][0], g["betas"][1],
                          g["eps"], self.step_dev, grad_scale)

    def load_named_state(self, module, m_by_name, v_by_name, step):
        """Load per-parameter Adam moments keyed by state_dict names (resume / parity tests)."""
        o = 0
        names = {id(p): k for k, p in module.named_parameters()}
        for p in self.flat.params:
            n = p.numel()
            k = names[id(p)]
            if k in m_by_name:
                self.m[o:o + n].copy_(m_by_name[k].reshape(-1))
                self.v[o:o + n].copy_(v_by_name[k].reshape(-1))
            o += n
        self.step_count = int(step)
        self.step_dev.fill_(int(step))

    def flat_state(self):
        return {"m": self.m, "v": self.v, "step": self.step_count, "lr": self.param_groups[0]["lr"]}

    def load_flat_state(self, sd):
        self.m.copy_(sd["m"])
        self.v.copy_(sd["v"])
        self.step_count = int(sd["step"])
        self.step_dev.fill_(self.step_count)
