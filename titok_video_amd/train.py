"""Generator training step on the HIP path (mirrors reference train.py:65-83 with the reconstruction term of
model/losses/loss_module.py:118: L1 between target and reconstruction, mean over clips of per-clip means).

    forward (tape) -> L1 loss -> backward (HIP kernels) -> [DP: count-weighted gradient all-reduce over RCCL]
    -> clip_grad_norm(max_grad_norm) -> optimizer.step()

`gan_training_step` adds the reference's discriminator step (train.py:86-107) on top of `model/losses/loss_module.py`'s
mirror; LPIPS is outside this path's scope (network-fetched weights).
"""
from __future__ import annotations

import os
from typing import List, Optional, Sequence

import torch

from . import dp
from .optim import HipAdamW, use_hip_adamw


class _L1LossFn(torch.autograd.Function):
    """Value and gradient of the L1 term for all clips in one HIP launch (ttv_l1_loss)."""

    @staticmethod
    def forward(ctx, n, *tensors):
        import ctypes as C
        from . import _lib
        recon = [t.contiguous() for t in tensors[:n]]
        target = [t.detach().to(recon[0].dtype).contiguous() for t in tensors[n:]]
        dev, dt = recon[0].device, recon[0].dtype
        loss = torch.zeros((), dtype=torch.float32, device=dev)
        grads = [torch.empty_like(r) for r in recon]
        sizes = (C.c_int32 * n)(*[int(r.numel()) for r in recon])
        _lib.check(_lib.lib().ttv_l1_loss(_lib.ptr_array(recon), _lib.ptr_array(target), _lib.ptr_array(grads), sizes, n,
                                          _lib.dtype_code(dt), loss.data_ptr(), _lib.stream_ptr(dev)), "ttv_l1_loss")
        ctx.grads, ctx.n = grads, n
        return loss

    @staticmethod
    def backward(ctx, gout):
        grads = ctx.grads
        torch._foreach_mul_(grads, gout.to(grads[0].dtype))
        return (None, *grads, *([None] * ctx.n))


def l1_reconstruction_loss(recon: Sequence[torch.Tensor], target: Sequence[torch.Tensor]) -> torch.Tensor:
    """Mean over clips of mean |target - recon| (loss_module.py:118 with per-clip tensors of different shapes).
    GPU tensors: one HIP kernel for value + gradient; CPU tensors (host-side tests of the DP logic): plain torch ops."""
    if recon[0].is_cuda:
        return _L1LossFn.apply(len(recon), *recon, *target)
    terms = [(r.float() - t.float()).abs().mean() for r, t in zip(recon, target)]
    return torch.stack(terms).mean()


def freeze_python_gc() -> None:
    """Host-side tuning for training loops (opt-in, process-wide): a full Python garbage collection walks every tracked
    object of the process (~170 k after importing torch: 15-25 ms on the GPU box) and the per-step allocations of an
    autograd step trigger one almost every step.  Freezing the objects that exist after set-up takes them out of the
    collector's view, so later collections only see what a step creates.  Measured: 29.5 -> 18 ms per training step."""
    import gc
    gc.collect()
    gc.freeze()


def host_cpu_share() -> int:
    """CPUs this process may actually use: the scheduler affinity, capped by the cgroup quota (a container sees every core of the
    host in os.cpu_count() - 256 on the GPU boxes - while its quota is 16)."""
    import os
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            with open(path) as f:
                parts = f.read().split()
            if path.endswith("cpu.max"):
                if parts[0] != "max":
                    n = min(n, max(1, int(parts[0]) // int(parts[1])))
            else:
                quota = int(parts[0])
                with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
                    period = int(f.read())
                if quota > 0:
                    n = min(n, max(1, quota // period))
            break
        except (OSError, ValueError, IndexError):
            continue
    return max(1, n)


def limit_host_threads(max_threads: int = 4) -> int:
    """Host-side tuning for training loops (opt-in, process-wide): torch sizes its intra-op thread pool by os.cpu_count() (128
    threads on a GPU box whose cgroup grants 16 CPUs).  The first CPU-side tensor operation above the parallel grain size wakes
    the whole pool and its workers then spin between parallel regions: the autograd thread, the loader's upload thread and the HIP
    runtime's own threads are starved (measured on BASELINE config #3, tools/train_host_diag.sh: hipLaunchKernel 5 -> 20 us, a
    training step 8.1 -> 33-40 ms although the GPU work is 8 ms).  A training process has no CPU-side tensor work worth more
    than a few threads; returns the thread count set."""
    n = max(1, min(int(max_threads), host_cpu_share(), torch.get_num_threads()))
    torch.set_num_threads(n)
    return n


def make_optimizer(model: torch.nn.Module, lr: float = 1e-4, beta1: float = 0.5, beta2: float = 0.96, weight_decay: float = 1e-4,
                   capturable: bool = False):
    """AdamW with the reference's hyper-parameters (configs/tiny.yaml:39-46, train.py:170-190).  capturable=True keeps the step
    counter on the device so that the step can be recorded into a HIP graph (GraphedTrainingStep)."""
    params = [p for p in model.parameters() if p.requires_grad]
    # parameters on the GPU: clip + AdamW as two HIP launches (optim.HipAdamW; TTV_HIP_ADAMW=0 keeps torch's multi-tensor kernels, and
    # so does capturable=True - the HIP step takes its step count from the host)
    if use_hip_adamw(params) and not capturable and os.environ.get("TTV_HIP_ADAMW", "1") != "0":
        return HipAdamW(params, lr=lr, betas=(beta1, beta2), weight_decay=weight_decay)
    fused = bool(params) and all(p.is_cuda for p in params)      # one multi-tensor kernel instead of a launch per parameter
    return torch.optim.AdamW(params, lr=lr, betas=(beta1, beta2), weight_decay=weight_decay, fused=fused, capturable=capturable and fused)


def _param_list(module) -> list:
    """module.parameters() as a list, cached on the module: the generator walks the module tree on every call (~0.1 ms for the tokenizer)
    and a training step that is launch-bound at the reference's batch sizes walked it three times.  A training loop does not register
    parameters between steps; one that does deletes module._ttv_param_list."""
    cached = module.__dict__.get("_ttv_param_list")
    if cached is None:
        cached = list(module.parameters())
        module.__dict__["_ttv_param_list"] = cached
    return cached


def clip_and_step(optimizer, params, max_grad_norm):
    """clip_grad_norm_(params, max_grad_norm) (skipped when max_grad_norm is falsy) + optimizer.step(); returns the gradient norm or None.
    HipAdamW does both in its own two launches (the norm is taken over the optimizer's parameters that have a gradient - the same set)."""
    if isinstance(optimizer, HipAdamW):
        if max_grad_norm:
            return optimizer.clip_and_step(max_grad_norm)
        optimizer.step()
        return None
    gnorm = torch.nn.utils.clip_grad_norm_(params, max_grad_norm) if max_grad_norm else None
    optimizer.step()
    return gnorm


def _reducer_for(model, group, overlap: bool):
    """The dp.GradReducer of a model's towers (created on first use, cached on the model) when torch.distributed is active and the
    gradients live on a GPU; None otherwise (single process, or `overlap=False`: the sequential reference path)."""
    rank, ws = dp.world()
    if ws == 1 or not overlap:
        return None
    dev = next(model.parameters()).device
    if dev.type != "cuda":
        return None
    red = model.__dict__.get("_dp_reducer")
    if red is None or red.group is not group:
        red = dp.GradReducer(dev, group)
        model.__dict__["_dp_reducer"] = red
    return red


def _towers(*modules):
    from .model.base.blocks import _Tower
    return [m for mod in modules for m in mod.modules() if isinstance(m, _Tower)]


def training_step(model, clips: List[torch.Tensor], token_counts, optimizer, max_grad_norm: float = 1.0,
                  target: Optional[List[torch.Tensor]] = None, group=None, overlap: bool = True):
    """One generator step on this rank's clips.  Returns (loss, grad_norm, indices).

    Under torch.distributed the gradients are reduced as sum(count * grad) / sum(count).  overlap=True (default): layer by layer
    on a communication stream while the backward is still running (dp.GradReducer); overlap=False: after the backward, in flat
    buckets (dp.allreduce_mean_by_count) - same values, kept as the reference for the overlapped path."""
    optimizer.zero_grad(set_to_none=True)
    red = _reducer_for(model, group, overlap)
    towers = _towers(model) if red is not None else []
    if red is not None:
        red.attach(*towers)
        red.begin_step(len(clips))
    try:
        recon, out = model(clips, token_counts)
        loss = l1_reconstruction_loss(recon, target if target is not None else clips)
        loss.backward()
    finally:
        if red is not None:
            red.detach(*towers)
    if red is not None:
        red.finish()                                   # joins the communication stream and delivers the towers' gradients
    params = [p for p in _param_list(model) if p.grad is not None]
    if red is not None:
        red.reduce_rest(params, len(clips))            # trainable parameters outside the towers (none in the reference's TiTok)
    else:
        dp.allreduce_mean_by_count([p.grad for p in params], len(clips), group=group)
    gnorm = clip_and_step(optimizer, params, max_grad_norm)
    return loss.detach(), gnorm, out["indices"]


class GraphedTrainingStep:
    """One generator training step (training_step: forward with tape, L1, backward, clip, AdamW) recorded ONCE into a HIP graph for a
    fixed batch shape and replayed per step ("HIP-graph capture of the training step").

    The tower entry points only enqueue kernels on the current stream (no allocation, no synchronisation inside the library), plans
    and weight packs are built during the warm-up steps, and the pack refresh after each optimizer step (parameter version -> new
    compute-dtype copies) is part of the captured sequence, so a replay is exact.  Constraints: one batch shape per instance (clip
    shapes and token counts fixed; ragged token-budget batches need one instance per shape signature), an optimizer created with
    capturable=True, single process (the overlapped DP reducer launches collectives from Python callbacks).

    Measured (tools/bench_train.py GRAPH=1): the replay runs at the speed of the eager step, 3.48 ms at 5 clips and 8.1 ms at 32 -
    with the Python garbage collector frozen the host issues the ~300 launches of a step faster than the GPU runs them (3.6 ms of
    kernel time at 5 clips: the small kernels are latency-bound at ~12 us each), so on this stack a graph buys nothing; what would
    is fewer launches.  Kept as a tested utility for hosts that ARE issue-bound.

        step = GraphedTrainingStep(model, opt, example_clips, counts)
        loss, grad_norm, indices = step(clips)          # copies the clips into the graph's input buffers, replays"""

    def __init__(self, model, optimizer, example_clips, token_counts, max_grad_norm: float = 1.0, warmup: int = 3):
        import copy
        dev = next(model.parameters()).device
        self.model, self.optimizer, self.counts = model, optimizer, list(token_counts)
        self.inputs = [c.detach().clone() for c in example_clips]
        params = [p for p in model.parameters()]
        saved_params = [p.detach().clone() for p in params]
        saved_opt = copy.deepcopy(optimizer.state_dict())          # empty state for a fresh optimizer
        self.graph = torch.cuda.CUDAGraph()
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):                   # warm-up on the capture stream: plans, packs, optimizer state, allocator pool
            for _ in range(warmup):
                training_step(model, self.inputs, self.counts, optimizer, max_grad_norm, overlap=False)
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize(dev)
        with torch.cuda.graph(self.graph, stream=side):
            self.loss, self.gnorm, self.indices = training_step(model, self.inputs, self.counts, optimizer, max_grad_norm, overlap=False)
        # the warm-up steps trained on the example batch: put parameters and optimizer state back IN PLACE (the graph holds their
        # addresses); a fresh optimizer goes back to zero moments and step 0
        with torch.no_grad():
            for p, q in zip(params, saved_params):
                p.copy_(q)
            old = saved_opt["state"]
            for i, p in enumerate(optimizer.param_groups[0]["params"]):
                st = optimizer.state.get(p, {})
                for k, v in st.items():
                    if torch.is_tensor(v):
                        if i in old and k in old[i]:
                            v.copy_(old[i][k])
                        else:
                            v.zero_()
        for mod in model.modules():                     # the packed copies were made from the warm-up weights
            if hasattr(mod, "invalidate_packs"):
                mod.invalidate_packs()

    def __call__(self, clips):
        torch._foreach_copy_(self.inputs, list(clips))
        self.graph.replay()
        return self.loss, self.gnorm, self.indices


def make_discriminator_optimizer(loss_module: torch.nn.Module, lr: float = 1e-4, disc_lr_ratio: float = 0.15, beta1: float = 0.5,
                                 beta2: float = 0.96, weight_decay: float = 1e-4):
    """AdamW over `loss_module.disc_model` at lr * disc_lr_ratio (reference train.py:195-201, configs/tiny.yaml:46)."""
    params = list(loss_module.disc_model.parameters())
    if use_hip_adamw(params) and os.environ.get("TTV_HIP_ADAMW", "1") != "0":           # as make_optimizer
        return HipAdamW(params, lr=lr * disc_lr_ratio, betas=(beta1, beta2), weight_decay=weight_decay)
    fused = bool(params) and all(p.is_cuda for p in params)
    return torch.optim.AdamW(params, lr=lr * disc_lr_ratio, betas=(beta1, beta2), weight_decay=weight_decay, fused=fused)


def gan_training_step(model, loss_module, clips: List[torch.Tensor], token_counts, opt_g, opt_d=None, max_grad_norm: float = 1.0,
                      group=None, overlap: bool = True):
    """One generator step followed by one discriminator step on this rank's clips (reference train.py:64-107).

    generator:      recon = model(clips); loss = loss_module(target=clips, recon=recon)  [L1 + disc_weight * relativistic GAN term
                    through the frozen discriminator]; backward; clip; opt_g.step()
    discriminator:  loss_module(target=clips, recon=recon, disc_forward=True)  [both detached inside: relativistic loss + R1/R2
                    finite-difference penalty + centering]; backward; clip; opt_d.step()      (skipped when disc_weight == 0)
    Under torch.distributed both gradient sets are reduced as sum(count * grad) / sum(count), as in `training_step` (overlap=True:
    slice by slice behind the backward on a communication stream, dp.GradReducer).
    Returns the merged {'gen/..', 'disc/..'} dictionary of detached scalars and the token indices."""
    red = _reducer_for(model, group, overlap)
    g_towers = _towers(model) if red is not None else []
    d_towers = _towers(loss_module) if red is not None else []
    opt_g.zero_grad(set_to_none=True)
    if red is not None:
        red.attach(*g_towers)          # the discriminator's towers stay detached: frozen in this step, no gradients of their own
        red.begin_step(len(clips))
    try:
        recon, out = model(clips, token_counts)
        loss, loss_dict = loss_module(target=clips, recon=recon)
        loss.backward()
    finally:
        if red is not None:
            red.detach(*g_towers)
    if red is not None:
        red.finish()
    g_params = [p for p in _param_list(model) if p.grad is not None]
    if red is not None:
        red.reduce_rest(g_params, len(clips))
    else:
        dp.allreduce_mean_by_count([p.grad for p in g_params], len(clips), group=group)
    clip_and_step(opt_g, g_params, max_grad_norm)
    if opt_d is not None and getattr(loss_module, "disc_weight", 0.0) > 0.0:
        opt_d.zero_grad(set_to_none=True)
        if red is not None:
            red.attach(*d_towers)
            red.begin_step(len(clips))
        try:
            d_loss, d_dict = loss_module(target=clips, recon=recon, disc_forward=True)
            loss_dict.update(d_dict)
            d_loss.backward()
        finally:
            if red is not None:
                red.detach(*d_towers)
        if red is not None:
            red.finish()
        d_params = [p for p in _param_list(loss_module.disc_model) if p.grad is not None]
        if red is not None:
            red.reduce_rest(d_params, len(clips))
        else:
            dp.allreduce_mean_by_count([p.grad for p in d_params], len(clips), group=group)
        clip_and_step(opt_d, d_params, max_grad_norm)
    return loss_dict, out["indices"]
