"""The training step of reference train.py:66-75 and its data-parallel extension.

    out  = model(x_history, x_inview, x_global)          train.py:69
    loss = model.loss(user_id, out, label)               train.py:71
    loss.backward(); optimizer.step(); optimizer.zero_grad()      train.py:73-75

Data parallel (new): one process per GPU, every rank takes B/N impressions, and ONE
all-reduce (RCCL over xGMI; gloo in the CPU tests) of a single flat fp32 gradient buffer sits between
backward and the Adam step.  BatchNorm statistics stay per replica.
"""
from __future__ import annotations

import torch
import torch.distributed as dist

from .config import Dims, model_config


def build_model(dims: Dims, user_num: int, state_dict=None, device="cuda"):
    """Construct a UserModel with the given dims (re-dimensions ``model_config`` the way the
    reference is re-dimensioned: dims are read at construction)."""
    from .modules import UserInvariantInterestModel, UserModel
    saved = dict(model_config)
    saved_defaults = UserInvariantInterestModel.__init__.__defaults__
    try:
        model_config["pca_vector"] = dims.pca_vector
        model_config["category_label_num"] = dims.category_label_num
        UserInvariantInterestModel.__init__.__defaults__ = (list(dims.embed_setting),)
        model = UserModel(user_num)
    finally:
        model_config.update(saved)
        UserInvariantInterestModel.__init__.__defaults__ = saved_defaults
    if state_dict is not None:
        model.load_state_dict({k: torch.as_tensor(v) for k, v in state_dict.items()}, strict=False)
    return model.to(device)


def make_optimizer(model, lr=1e-3):
    """train.py:48 -- Adam(lr, weight_decay=1e-5), L2 folded into the gradient."""
    return torch.optim.Adam(model.parameters(), lr=lr, weight_decay=1e-5)


class FlatGradReducer:
    """Averages the gradients of ``params`` across ranks with ONE all-reduce of a flat fp32 buffer."""

    def __init__(self, params, group=None):
        self.params = [p for p in params if p.requires_grad]
        self.group = group
        n = sum(p.numel() for p in self.params)
        dev = self.params[0].device if self.params else "cpu"
        self.flat = torch.zeros(n, dtype=torch.float32, device=dev)
        self.views, off = [], 0
        for p in self.params:
            self.views.append(self.flat[off:off + p.numel()].view_as(p))
            off += p.numel()

    @property
    def nbytes(self):
        return self.flat.numel() * 4

    def reduce(self):
        world = dist.get_world_size(self.group) if dist.is_initialized() else 1
        if world == 1:
            return
        for p, v in zip(self.params, self.views):
            if p.grad is None:
                v.zero_()
            else:
                v.copy_(p.grad)
        dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=self.group)
        self.flat.mul_(1.0 / world)
        for p, v in zip(self.params, self.views):
            if p.grad is None:
                p.grad = v.clone()
            else:
                p.grad.copy_(v)


def train_step(model, optimizer, batch, reducer: FlatGradReducer | None = None, alpha=0.95):
    """One step of train.py:69-75 on device-resident tensors; returns (loss, out) detached."""
    out = model(batch["x_history"], batch["x_target"], batch["x_global"])
    loss = model.loss(batch["user_id"], out, batch["label"], alpha)
    loss.backward()
    if reducer is not None:
        reducer.reduce()
    optimizer.step()
    optimizer.zero_grad(set_to_none=False)
    return loss.detach(), out.detach()


def shard_batch(batch, rank, world):
    """Rank ``rank`` of ``world`` takes a contiguous 1/world slice of the impressions."""
    B = batch["user_id"].shape[0]
    if B % world:
        raise ValueError(f"batch {B} does not divide over {world} ranks")
    lo, hi = rank * (B // world), (rank + 1) * (B // world)
    return {k: (v[lo:hi] if hasattr(v, "shape") and v.ndim > 0 and v.shape[0] == B else v) for k, v in batch.items()}


def batch_to_device(batch_np, device="cuda", dtype=None):
    out = {}
    for k, v in batch_np.items():
        if k in ("user_num",):
            continue
        t = torch.as_tensor(v)
        if dtype is not None and t.is_floating_point():
            t = t.to(dtype)
        out[k] = t.to(device)
    return out
