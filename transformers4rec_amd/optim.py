"""Flat parameter / gradient buffers and the fused Adam step.

The reference trains with torch.optim.Adam (Model.fit, transformers4rec/torch/model/base.py:
669-718) and lets torch DDP bucket the gradients.  On MI355X every parameter lives in one of
two flat fp32 buffers -- "dense" (everything but the embedding tables) and "tables" -- so that
  * the Adam update is ONE launch per buffer (csrc/elementwise.hip adam_kernel),
  * the data-parallel exchange is ONE RCCL all-reduce per buffer (distributed.py),
  * XLNet's q,k,v weights are adjacent and go through one batched GEMM.
`nn.Parameter.data` / `.grad` become views into the flat buffers; names and shapes are untouched.
"""
import torch

from . import ops


class FlatParams:
    def __init__(self, named_params, align=4):
        params = []
        seen = set()
        for name, p in named_params:
            if id(p) in seen or not p.requires_grad:
                continue
            seen.add(id(p))
            params.append((name, p))
        if not params:
            raise ValueError("no parameters to flatten")
        dev = params[0][1].device
        offs, n = [], 0
        for _, p in params:
            offs.append(n)
            n += (p.numel() + align - 1) // align * align
        self.data = torch.zeros(n, device=dev, dtype=torch.float32)
        self.grad = torch.zeros(n, device=dev, dtype=torch.float32)
        self.entries = []
        for (name, p), o in zip(params, offs):
            view = self.data[o: o + p.numel()].view(p.shape)
            view.copy_(p.data)
            p.data = view
            p.grad = self.grad[o: o + p.numel()].view(p.shape)
            self.entries.append((name, p, o))
        self.numel = n

    def ensure_grads(self):
        """Re-attach .grad views (e.g. after zero_grad(set_to_none=True))."""
        for _, p, o in self.entries:
            if p.grad is None or p.grad.data_ptr() != self.grad.data_ptr() + 4 * o:
                g = self.grad[o: o + p.numel()].view(p.shape)
                if p.grad is not None:
                    g.copy_(p.grad)
                p.grad = g


def flatten_model(model):
    """-> (dense FlatParams, tables FlatParams or None).  Embedding tables (and an untied output
    layer) form their own bucket: they dominate the bytes."""
    dense, tables = [], []
    for name, p in model.named_parameters():
        is_table = (".embedding_tables." in name and name.endswith(".weight") and p.ndim == 2
                    and "continuous_module" not in name) or name.endswith("output_layer")
        (tables if is_table else dense).append((name, p))
    return FlatParams(dense), (FlatParams(tables) if tables else None)


class FusedAdam:
    """torch.optim.Adam semantics over FlatParams buffers; zeroes the gradients in the same pass."""

    def __init__(self, flats, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        self.flats = [f for f in flats if f is not None]
        self.lr, self.betas, self.eps, self.weight_decay = lr, betas, eps, weight_decay
        self.state = [(torch.zeros_like(f.data), torch.zeros_like(f.data)) for f in self.flats]
        self.step_count = 0
        # per bucket: (parameter, offset, partial-maxima buffer) of its largest 2-D table on the GPU, or None
        self._amax_targets = []
        for f in self.flats:
            tabs = [(p.numel(), p, o) for name, p, o in f.entries if p.ndim == 2 and ".embedding_tables." in name and "continuous_module" not in name and f.data.is_cuda]
            if tabs:
                _, p, o = max(tabs, key=lambda t: t[0])
                self._amax_targets.append((p, o, torch.zeros(1024, device=f.data.device, dtype=torch.float32)))
            else:
                self._amax_targets.append(None)

    @staticmethod
    def _join():
        # weight-gradient streams a failed backward pass left unjoined (transformer.py; no-op normally)
        from .transformer import _join_weight_gradient_streams
        _join_weight_gradient_streams()

    def step(self, grad_scale=1.0):
        self._join()
        self.step_count += 1
        for k, (f, (m, v)) in enumerate(zip(self.flats, self.state)):
            f.ensure_grads()
            tgt = self._amax_targets[k]
            if tgt is None:
                ops.adam_step_(f.data, f.grad, m, v, self.step_count, self.lr, self.betas, self.eps,
                               self.weight_decay, grad_scale, zero_grad=True)
                continue
            # the bucket's largest table (the tied item table of a next-item model): its maximum after the update comes out
            # of this launch, for the head of the NEXT step (ops.w_amax_of checks that nothing wrote the table in between)
            p, lo, part = tgt
            n = ops.adam_step_amax_(f.data, f.grad, m, v, self.step_count, lo, lo + p.numel(), part, self.lr, self.betas, self.eps,
                                    self.weight_decay, grad_scale, zero_grad=True)
            p._t4r_w_amax = (part, n, p.data_ptr(), p._version, f.data, f.data._version)

    def zero_grad(self):
        self._join()
        for f in self.flats:
            f.grad.zero_()
