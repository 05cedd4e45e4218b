"""AdamW + gradient-norm clipping of the training loop as two HIP launches (`ttv_opt_grad_sumsq`, `ttv_opt_adamw_step`).

The reference steps `torch.optim.AdamW` after `clip_gradients` (train.py:76-77, :183-190).  On MI355X torch's multi-tensor path costs
seven launches and ~230 us per step for the tiny tokenizer's 7 M parameters; `HipAdamW` keeps torch's interface, hyper-parameters and
state layout (`step`, `exp_avg`, `exp_avg_sq` per parameter: a `state_dict()` loads into `torch.optim.AdamW` and back) and does the
arithmetic in csrc/ttv_train.hip.  GPU only: there is no host fallback (parameters on the CPU raise)."""
import math
from typing import Iterable, List, Optional

import torch

from . import _lib

_CHUNK = 8192          # OPT_CHUNK of csrc/ttv_train.hip
_SLOTS = 4             # host / device table buffers in rotation: a slot is reused only after the step that read it has finished


class _Tables:
    """Pinned host + device buffers for the per-step pointer tables (gradients are fresh tensors every step, so the table is rebuilt and
    uploaded each time: ~4 KB).  A slot's event is recorded behind the kernels that read it and waited for before the slot is rewritten."""

    def __init__(self, device):
        self.device = device
        self.slots = []
        self.next = 0

    def take(self, n_entries: int, n_chunks: int):
        if len(self.slots) < _SLOTS:
            self.slots.append(None)
        i = self.next
        self.next = (self.next + 1) % _SLOTS
        need = n_entries * 5 + n_chunks           # int64 words: 5 per entry, one per chunk (two int32)
        s = self.slots[i]
        if s is None or s["host"].numel() < need:
            cap = max(need, 4096)
            s = {"host": torch.empty(cap, dtype=torch.int64).pin_memory(), "dev": torch.empty(cap, dtype=torch.int64, device=self.device),
                 "event": None}
            self.slots[i] = s
        elif s["event"] is not None:
            s["event"].synchronize()
        return s


class HipAdamW(torch.optim.Optimizer):
    """torch.optim.AdamW (amsgrad=False, maximize=False) on the HIP kernels.  `step()` = AdamW; `clip_and_step(max_norm)` = clip_grad_norm_
    over every parameter of every group followed by AdamW, and returns the gradient norm (a device scalar, no synchronisation).  The clip
    factor is applied inside the update: `p.grad` keeps its unclipped values."""

    def __init__(self, params: Iterable, lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8, weight_decay: float = 1e-2):
        if lr < 0 or eps < 0 or weight_decay < 0 or not (0 <= betas[0] < 1) or not (0 <= betas[1] < 1):
            raise ValueError("HipAdamW: invalid hyper-parameter")
        super().__init__(params, dict(lr=lr, betas=tuple(betas), eps=eps, weight_decay=weight_decay))
        self._tables = None
        self._partials = None
        self._norm = None
        self._static = {}

    # -- state ----------------------------------------------------------------------------------------------------------------------
    def _state_for(self, p):
        st = self.state[p]
        if not st:
            st["step"] = torch.tensor(0.0, dtype=torch.float32)
            st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
            st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
        return st

    @staticmethod
    def _step_value(st) -> float:
        s = st["step"]
        return float(s.item()) if torch.is_tensor(s) else float(s)

    # -- the step -------------------------------------------------------------------------------------------------------------------
    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        self._run(None)
        return loss

    @torch.no_grad()
    def clip_and_step(self, max_norm: float) -> torch.Tensor:
        return self._run(float(max_norm))

    def _run(self, max_norm: Optional[float]):
        # buckets: (group, dtype) -> parameters with a gradient
        buckets = []
        for g in self.param_groups:
            by_dt = {}
            for p in g["params"]:
                if p.grad is None:
                    continue
                if not p.is_cuda:
                    raise RuntimeError("HipAdamW: parameter on " + str(p.device) + "; the optimizer kernels run on the GPU only")
                if p.grad.is_sparse or p.grad.dtype != p.dtype or not p.is_contiguous() or not p.grad.is_contiguous():
                    raise RuntimeError("HipAdamW: dense contiguous gradients of the parameter's dtype only")
                by_dt.setdefault(p.dtype, []).append(p)
            for dt, ps in by_dt.items():
                buckets.append((g, dt, ps))
        clip = max_norm is not None
        if not buckets:
            return torch.zeros((), device="cuda") if clip else None
        dev = buckets[0][2][0].device
        if any(p.device != dev for _, _, ps in buckets for p in ps):
            raise RuntimeError("HipAdamW: all parameters must live on one device")
        lib = _lib.lib()
        stream = _lib.stream_ptr(dev)
        if self._tables is None or self._tables.device != dev:
            self._tables = _Tables(dev)
            self._norm = torch.zeros(1, dtype=torch.float32, device=dev)
            self._static = {}
        # Host tables.  Element counts, chunk lists and offsets of a parameter list are built once; the pointers every step.
        for _, _, ps in buckets:
            for p in ps:
                self._state_for(p)
        key = tuple((id(g), dt, tuple((id(p), p.numel()) for p in ps)) for g, dt, ps in buckets)
        st = self._static.get(key)
        if st is None:
            words, chunk_words, layout = [], [], []
            e_off = c_off = 0
            for g, dt, ps in buckets:
                c0 = c_off
                for i, p in enumerate(ps):
                    words += [0, 0, 0, 0, p.numel()]
                    for first in range(0, p.numel(), _CHUNK):
                        chunk_words.append(i | (first << 32))      # int2 {entry index within the bucket's table, first element}
                        c_off += 1
                layout.append((e_off, len(ps), c0, c_off - c0))
                e_off += len(ps)
            st = {"words": torch.tensor(words + chunk_words, dtype=torch.int64), "n_words": len(words), "layout": layout, "n_chunks": c_off}
            self._static = {key: st}          # one live parameter list at a time
        tmpl, n_words, layout, n_chunks = st["words"], st["n_words"], st["layout"], st["n_chunks"]
        # the four pointer columns are read afresh (a load_state_dict() or a .data assignment may have replaced a tensor)
        flat = [p for _, _, ps in buckets for p in ps]
        ptrs = []
        for p in flat:
            sp = self.state[p]
            if sp["exp_avg"].dtype != p.dtype or sp["exp_avg"].device != p.device:          # a loaded state_dict: bring it to the parameter
                sp["exp_avg"] = sp["exp_avg"].to(device=p.device, dtype=p.dtype)
                sp["exp_avg_sq"] = sp["exp_avg_sq"].to(device=p.device, dtype=p.dtype)
            ptrs += [p.data_ptr(), p.grad.data_ptr(), sp["exp_avg"].data_ptr(), sp["exp_avg_sq"].data_ptr()]
        tmpl[:n_words].view(-1, 5)[:, :4] = torch.tensor(ptrs, dtype=torch.int64).view(-1, 4)
        slot = self._tables.take(n_words // 5, n_chunks)
        total = tmpl.numel()
        slot["host"][:total].copy_(tmpl)
        dev_buf = slot["dev"]
        dev_buf[:total].copy_(slot["host"][:total], non_blocking=True)
        base = dev_buf.data_ptr()
        chunks_base = base + 8 * n_words
        if self._partials is None or self._partials.numel() < max(n_chunks, 1) or self._partials.device != dev:
            self._partials = torch.empty(max(n_chunks, 1024), dtype=torch.float32, device=dev)
        if clip:
            for (g, dt, ps), (eo, ne, co, nc) in zip(buckets, layout):
                _lib.check(lib.ttv_opt_grad_sumsq(base + 40 * eo, chunks_base + 8 * co, nc, _lib.dtype_code(dt),
                                                  self._partials.data_ptr() + 4 * co, stream), "opt_grad_sumsq")
        for (g, dt, ps), (eo, ne, co, nc) in zip(buckets, layout):
            # torch keeps a step count per parameter; they differ only when parameters join a group late, which this path does not support
            steps = {self._step_value(self.state[p]) for p in ps}
            if len(steps) != 1:
                raise RuntimeError("HipAdamW: parameters of one group with different step counts are not supported")
            t = steps.pop() + 1.0
            b1, b2 = g["betas"]
            _lib.check(lib.ttv_opt_adamw_step(base + 40 * eo, chunks_base + 8 * co, nc, _lib.dtype_code(dt), self._partials.data_ptr(),
                                              n_chunks if clip else 0, float(g["lr"]), float(b1), float(b2), float(g["eps"]),
                                              float(g["weight_decay"]), 1.0 - b1 ** t, math.sqrt(1.0 - b2 ** t),
                                              float(max_norm) if clip else 0.0, self._norm.data_ptr() if clip else None, stream),
                       "opt_adamw_step")
            for p in ps:          # one tensor PER parameter, as torch keeps them (its single-tensor step increments each in place)
                sp = self.state[p]
                if torch.is_tensor(sp["step"]) and sp["step"].device.type == "cpu":
                    sp["step"].fill_(t)
                else:
                    sp["step"] = torch.tensor(t, dtype=torch.float32)
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(dev))
        slot["event"] = ev
        return self._norm[0].clone() if clip else None


def use_hip_adamw(params: List[torch.nn.Parameter]) -> bool:
    """The HIP optimizer applies when every parameter is a dense tensor on one GPU."""
    return bool(params) and all(p.is_cuda for p in params) and len({p.device for p in params}) == 1
