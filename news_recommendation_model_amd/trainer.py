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


def build_model(dims: Dims, user_num: int, state_dict=None, device="cuda", attention_mma=None):
    """Construct a UserModel with the given dims (re-dimensions ``model_config`` the way the
    reference is re-dimensioned: dims are read at construction).  ``attention_mma`` ('f32' | 'bf16') pins the
    arithmetic of both attentions' contraction on this model (None: the process default, fp32)."""
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
    if attention_mma is not None:
        from . import ops
        ops.resolve_mma(attention_mma)                        # validates the name
        model.invariant_interest_model.label_attention.mma = attention_mma
        model.invariant_interest_model.text_img_attention.mma = attention_mma
    return model.to(device)


def make_optimizer(model, lr=1e-3):
    """train.py:48 -- Adam(lr, weight_decay=1e-5), L2 folded into the gradient."""
    return torch.optim.Adam(model.parameters(), lr=lr, weight_decay=1e-5)


class FlatAdam:
    """train.py:48's Adam(lr, weight_decay=1e-5) over ONE flat fp32 buffer.

    Parameters are re-seated as views into one flat buffer (values are preserved, the Module's ``state_dict`` keeps
    working).  ``loss.backward()`` leaves every parameter's gradient where autograd produced it (``p.grad`` starts as
    None, so AccumulateGrad adopts the tensor instead of launching an ``add_`` per parameter); ``collect_grads()`` then
    gathers all of them into ``flat_grad`` with ONE launch (C ABI ``nrm_gather_flat``), the data-parallel all-reduce runs on
    that buffer without a copy, and ``step()`` is one fused Adam launch over all weights."""

    def __init__(self, model, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-5):
        import ctypes
        from . import native
        native.load()
        self.params = [p for p in model.parameters() if p.requires_grad]
        if not self.params or not self.params[0].is_cuda:
            raise RuntimeError("FlatAdam needs the model on an MI355X device (no CPU path)")
        dev = self.params[0].device
        offs, n = [], 0
        for p in self.params:
            offs.append(n)
            n += (p.numel() + 3) // 4 * 4                      # every parameter starts 16-byte aligned
        self.n = n
        self.offsets = offs
        self.flat_param = torch.zeros(n, dtype=torch.float32, device=dev)
        self.flat_grad = torch.zeros(n, dtype=torch.float32, device=dev)
        self.exp_avg = torch.zeros(n, dtype=torch.float32, device=dev)
        self.exp_avg_sq = torch.zeros(n, dtype=torch.float32, device=dev)
        for p, o in zip(self.params, offs):
            view = self.flat_param[o:o + p.numel()].view_as(p)
            view.copy_(p.data)
            p.data = view
            p.grad = None
        k = len(self.params)
        self._srcs = (ctypes.c_void_p * k)()
        self._offs = (ctypes.c_long * k)(*offs)
        self._cnts = (ctypes.c_long * k)(*[p.numel() for p in self.params])
        self._collected = False
        self.lr, self.betas, self.eps, self.weight_decay = lr, betas, eps, weight_decay
        # {step, 1-b1^step, sqrt(1-b2^step), -} lives on the device: the step is graph-capturable
        self.state = torch.zeros(4, dtype=torch.float32, device=dev)
        self.param_groups = [{"lr": lr}]                       # what train.py:57 reads
        self.collective_events = None                          # a list: all_reduce_grads appends an event pair per call (bench.py)

    @property
    def nbytes(self):
        return self.n * 4

    @property
    def steps(self):
        return int(self.state[0].item())

    def grad_view(self, p_index):
        """The slot of parameter ``p_index`` in ``flat_grad`` (valid after ``collect_grads()``)."""
        p, o = self.params[p_index], self.offsets[p_index]
        return self.flat_grad[o:o + p.numel()].view_as(p)

    def collect_grads(self):
        """All ``p.grad`` -> ``flat_grad`` (a parameter without a gradient contributes zeros), one launch; the per-parameter
        gradient tensors are released.  Idempotent until the next ``step()`` / ``zero_grad()``."""
        if self._collected:
            return
        from . import native, ops
        ops.verify_deferred_targets(self.params)   # a deferred reduction whose buffer autograd copied away would be lost: raise
        ops.flush_slab_reductions()                # weight gradients whose slab reduction train_step deferred become valid here
        keep = []
        for i, p in enumerate(self.params):
            g = p.grad
            if g is None:
                self._srcs[i] = None
                continue
            if g.dtype != torch.float32 or not g.is_contiguous():
                g = g.to(torch.float32).contiguous()
            keep.append(g)
            self._srcs[i] = g.data_ptr()
        native.call("nrm_gather_flat", self._srcs, self._offs, self._cnts, len(self.params), native.ptr(self.flat_grad), self.n,
                    native.stream_ptr())
        for p in self.params:
            p.grad = None
        self._collected = True

    def step(self, zero_grad=True):
        from . import ops
        self.collect_grads()
        ops.adam_step(self.flat_param, self.flat_grad, self.exp_avg, self.exp_avg_sq, self.state,
                      float(self.param_groups[0]["lr"]), float(self.betas[0]), float(self.betas[1]), float(self.eps),
                      float(self.weight_decay), bool(zero_grad))          # torch.ops.nrm.adam_step -> nrm_adam_step_dev
        # the weights moved (through a raw pointer: no autograd version bump): every cached packed GEMM operand is
        # refreshed by ONE launch instead of one per GEMM call of the next step
        ops.repack_persistent(self.params)
        self._collected = False

    def zero_grad(self, set_to_none=False):
        for p in self.params:
            p.grad = None
        self.flat_grad.zero_()
        self._collected = False

    def all_reduce_grads(self, group=None):
        """The ONE collective of a data-parallel step: sum the flat gradient over ranks, then average."""
        self.collect_grads()
        if not dist.is_initialized():
            return
        # issued whenever a process group exists, also for a group of one (bench.py's NRM_DIST_WORLD1 rehearsal of RCCL)
        world = dist.get_world_size(group)
        timed = self.collective_events is not None and not torch.cuda.is_current_stream_capturing()
        if timed:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        if dist.get_backend(group) == "nccl":
            dist.all_reduce(self.flat_grad, op=dist.ReduceOp.AVG, group=group)      # RCCL averages in the collective
        else:
            dist.all_reduce(self.flat_grad, op=dist.ReduceOp.SUM, group=group)
            if world > 1:
                self.flat_grad.mul_(1.0 / world)
        if timed:
            e1.record()
            self.collective_events.append((e0, e1))


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


class IndexErrorWatch:
    """Out-of-range ids without a host sync.  The kernels clamp a bad table index / user id and set a flag (the reference raises
    IndexError at the offending batch: F.embedding at models/user_invariant_interest_model.py:59, delta[id] at
    models/user_model.py:40).  The flag lives in pinned host memory the device writes directly (ops.index_error_flag), so
    ``before_step()`` just reads it: the error surfaces at the start of the step after the offending batch when the device keeps
    up with the host, and ``max_lag`` + 1 steps after it at the latest (``after_step()`` records an event per step and
    ``before_step()`` waits for the one that is ``max_lag`` steps old) -- not at the epoch's end, and at no cost per step beyond
    one event record.

    What this is NOT: the reference raises BEFORE any update; here the offending batch's step (computed with the clamped ids) and
    up to ``max_lag`` following steps have already been applied to the weights, both Adam moments and ``delta`` when the
    exception arrives -- the exception says so.  A caller that must not train on a bad batch uses the strict mode instead
    (``train_step(..., strict_ids=True)`` / ``validate_batch_ids``: ids checked on the device before the step is enqueued, one
    host synchronisation per step).  When the flag is found raised, every step still in flight is waited for BEFORE the flag is
    cleared, so a step enqueued earlier cannot raise it again for an unrelated later batch."""

    def __init__(self, device, max_lag=2):
        from . import ops
        self.flag = ops.index_error_flag(device)
        self.device = torch.device(device)
        self.max_lag = max_lag
        self.pending = []                                   # one event per enqueued step, oldest first

    def after_step(self):
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(self.device))
        self.pending.append(ev)

    def before_step(self):
        while self.pending and (len(self.pending) > self.max_lag or self.pending[0].query()):
            self.pending.pop(0).synchronize()
        if int(self.flag[0]):
            in_flight = len(self.pending)
            while self.pending:                             # drain: steps still in flight may set the flag again
                self.pending.pop(0).synchronize()
            self.flag.zero_()
            raise IndexError("index out of range (a category / type / time table index of a packed feature row, or a user "
                             f"id outside delta) in a batch of one of the last {in_flight + 1} steps; those steps ran with the "
                             "offending ids clamped and HAVE ALREADY UPDATED the weights, the Adam moments and delta (the "
                             "reference raises before any update: use train_step(..., strict_ids=True) for that behaviour)")


def validate_batch_ids(model, batch):
    """The reference's IndexError BEFORE anything is enqueued (strict mode of train_step): every table index of the packed rows
    (time4 | text_img[P] | category | sub-categories | sentiment | type ...: models/user_invariant_interest_model.py:14-22,58-71)
    and every user id (models/user_model.py:40; negative ids count from the end, as torch indexing does) is range-checked on the
    device; ONE host synchronisation.  The asynchronous default is IndexErrorWatch."""
    if torch.cuda.is_current_stream_capturing():
        # the check ends in ONE host read of a device flag: inside a stream capture that is an error that can take the process
        # down (global capture mode).  A captured step cannot raise per replay anyway: validate before capturing / replaying.
        raise RuntimeError("validate_batch_ids / train_step(strict_ids=True) synchronises with the host and cannot run inside a "
                           "stream capture; validate the batch before the capture (or rely on IndexErrorWatch around replays)")
    inv = model.invariant_interest_model
    d = inv._dims
    P, ns = d.pca_vector, d.n_subcat
    dev = batch["x_history"].device
    tables = torch.tensor([inv.year_embedding[0].num_embeddings, inv.month_embedding[0].num_embeddings,
                           inv.day_embedding[0].num_embeddings, inv.hour_embedding[0].num_embeddings], device=dev)
    n_cat, n_type = inv.category_embedding[0].num_embeddings, inv.type_embedding[0].num_embeddings
    flags = []
    for x in (batch["x_history"], batch["x_target"]):
        if x.numel() == 0:
            continue
        t4 = x[..., :4].long()                                           # the four time-table indices of a row, one comparison
        flags.append(((t4 < 0) | (t4 >= tables)).any())                  # (F.embedding refuses negative indices too)
        ids = x[..., 4 + P:4 + P + 1 + ns].long()                     # category and its sub-category slots share one table
        flags.append(((ids < 0) | (ids >= n_cat)).any())
        typ = x[..., 4 + P + 1 + ns + d.n_sentiment].long()
        flags.append(((typ < 0) | (typ >= n_type)).any())
    n = model.delta.numel()
    uid = batch["user_id"].long()
    flags.append(((uid < -n) | (uid >= n)).any())
    if bool(torch.stack(flags).any()):                                 # the one synchronisation
        raise IndexError("index out of range in self (a category / type / time table index of a packed feature row, or a user id "
                         "outside delta); nothing was enqueued, no weight was updated")


_watches = {}


def _index_watch(device):
    device = torch.device(device)
    key = (device.type, device.index if device.index is not None else torch.cuda.current_device())
    w = _watches.get(key)
    if w is None:
        w = _watches[key] = IndexErrorWatch(device)
    return w


def train_step(model, optimizer, batch, reducer: FlatGradReducer | None = None, alpha=0.95, defer_reductions=True,
               strict_ids=False):
    """One step of train.py:69-75 on device-resident tensors; returns (loss, out) detached.  An out-of-range id of an EARLIER
    step raises IndexError here, before this step's work is enqueued (IndexErrorWatch: asynchronous, the offending step has
    been applied by then).  ``strict_ids=True``: the ids of THIS batch are checked first (validate_batch_ids, one host
    synchronisation) and the IndexError is raised before anything is enqueued -- the reference's behaviour."""
    watch = _index_watch(batch["x_history"].device) if not torch.cuda.is_current_stream_capturing() else None
    if watch is not None:
        watch.before_step()
    if strict_ids:
        validate_batch_ids(model, batch)
    out = model(batch["x_history"], batch["x_target"], batch["x_global"])
    loss = model.loss(batch["user_id"], out, batch["label"], alpha)
    if isinstance(optimizer, FlatAdam):
        from . import ops
        ops.begin_step()                              # (stream ordering of the two attentions' backward contractions: ops._chain)
        # nobody reads a weight gradient between backward() and collect_grads(): the step's slab reductions (one per weight
        # gradient) are recorded during backward and run as ONE launch when FlatAdam gathers the gradients.  A gradient that
        # exists already would be accumulated into before its reduction ran: then (and on request) reductions run at once.
        defer = defer_reductions and all(p.grad is None for p in optimizer.params)
        seed = ops.unit_grad(loss) if loss.dim() == 0 and loss.dtype == torch.float32 else None     # (see ops.unit_grad)
        if defer:
            with ops.deferred_slab_reductions():
                loss.backward(seed)
        else:
            loss.backward(seed)
        optimizer.all_reduce_grads()
        optimizer.step(zero_grad=True)                # Adam + zero_grad fused in one launch
    else:
        loss.backward()
        if reducer is not None:
            reducer.reduce()
        optimizer.step()
        optimizer.zero_grad(set_to_none=False)
    if watch is not None:
        watch.after_step()
    return loss.detach(), out.detach()


class GraphedTrainStep:
    """The whole train.py:69-75 step captured once into a HIP graph and replayed (static shapes, static input
    buffers).  A replay is a single launch instead of ~600: 2-13 % faster than the eager step depending on the shape
    (reference default dimensions: 3.1 -> 2.7 ms).  New data is copied INTO the tensors
    of ``batch`` before ``replay()``; ``loss`` / ``out`` are overwritten in place by every replay."""

    def __init__(self, model, optimizer, batch, alpha=0.95, warmup=3, pool=None):
        """``pool``: the memory pool of another GraphedTrainStep (``other.graph.pool()``) -- several captured steps, e.g. one per
        resident input batch, then share their intermediates' memory (they must be replayed one at a time, in any order)."""
        if not isinstance(optimizer, FlatAdam):
            raise TypeError("GraphedTrainStep needs trainer.FlatAdam (its step counter lives on the device)")
        from . import native
        if native.kernel_events is not None:
            raise RuntimeError("per-kernel event timing cannot be recorded inside a graph capture")
        self.batch = batch
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):                       # allocator / lazy-init warm-up outside the capture
            for _ in range(warmup):
                train_step(model, optimizer, batch, None, alpha)
        torch.cuda.current_stream().wait_stream(side)
        self.graph = torch.cuda.CUDAGraph()
        # With a process group the captured step contains the RCCL all-reduce, and ProcessGroupNCCL's watchdog THREAD keeps
        # polling events (hipEventQuery) of earlier collectives: under the default "global" capture mode any such call from
        # another thread while this one captures is an error that takes the process down ("operation not permitted when stream
        # is capturing" -- seen once in five runs of the one-rank RCCL rehearsal).  Thread-local capture mode restricts only the
        # capturing thread; the device is drained first so that no collective of the warm-up steps is still in flight.
        mode = "global"
        if dist.is_available() and dist.is_initialized():
            torch.cuda.synchronize()
            mode = "thread_local"
        with torch.cuda.graph(self.graph, pool=pool, capture_error_mode=mode):
            self.loss, self.out = train_step(model, optimizer, batch, None, alpha)

    def replay(self):
        watch = _index_watch(self.batch["x_history"].device)
        watch.before_step()                                  # an out-of-range id of an earlier replay raises IndexError here
        self.graph.replay()
        watch.after_step()
        return self.loss, self.out


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


class BatchPrefetcher:
    """Host -> HBM staging of DataLoader batches on a side stream, one batch ahead of the step.

    The reference moves every field with ``.to(device)`` inside the step (train.py:66-68), float64 as the DataLoader
    yields it; at C3 that is 272 MB per step.  Here the next batch is copied from pinned host memory into one of two
    preallocated device buffer sets on a copy stream while the current step runs; the compute stream waits on the
    copy's event, and a buffer is only overwritten after the step that read it has been enqueued (its event)."""

    def __init__(self, batches, device="cuda", pin=True):
        self.device = torch.device(device)
        self.batches = iter(batches)
        self.pin = pin
        self.copy_stream = torch.cuda.Stream(device=self.device)
        self.bufs = [None, None]
        self.ready = [torch.cuda.Event(), torch.cuda.Event()]      # copy finished
        self.free = [None, None]                                    # step that consumed the buffer enqueued
        self.slot = 0
        self._pending = None
        self._stage()

    def _host(self, batch):
        out = {}
        for k, v in batch.items():
            if k == "user_num":
                continue
            t = torch.as_tensor(v)
            out[k] = t.pin_memory() if self.pin and not t.is_pinned() else t
        return out

    def _stage(self):
        try:
            host = self._host(next(self.batches))
        except StopIteration:
            self._pending = None
            return
        i = self.slot
        if self.bufs[i] is None or any(self.bufs[i][k].shape != v.shape or self.bufs[i][k].dtype != v.dtype
                                       for k, v in host.items()):
            self.bufs[i] = {k: torch.empty(v.shape, dtype=v.dtype, device=self.device) for k, v in host.items()}
            self.free[i] = None
        with torch.cuda.stream(self.copy_stream):
            if self.free[i] is not None:
                self.copy_stream.wait_event(self.free[i])
            for k, v in host.items():
                self.bufs[i][k].copy_(v, non_blocking=True)
            self.ready[i].record(self.copy_stream)
        self._pending = (i, host)                                   # keep the pinned tensors alive until consumed

    def __iter__(self):
        return self

    def __next__(self):
        if self._pending is None:
            raise StopIteration
        i, _host = self._pending
        torch.cuda.current_stream(self.device).wait_event(self.ready[i])
        batch = self.bufs[i]
        self.slot = 1 - i
        self._stage()                                               # next copy overlaps the step about to be enqueued
        return batch, i

    def release(self, slot):
        """Call after the step that read buffer set ``slot`` has been enqueued on the current stream."""
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(self.device))
        self.free[slot] = ev


def train_epochs(model, optimizer, make_loader, epochs, device="cuda", ckpt_path=None, on_batch=None):
    """The epoch loop of reference train.py:52-100 around ``train_step``: every epoch re-reads ``param_groups[0]['lr']``
    (:57), steps through ``make_loader()`` (an iterable of host batches, dict fields as in ``synth.make_batch``), keeps
    the running loss and per-impression AUC averages the reference prints (:77-88, AUC on the device instead of a
    host sklearn loop), and saves the state_dict without ``delta`` (:95-97).  The reference never steps its scheduler
    (:99-100), so none is taken here.  Batches are staged through ``BatchPrefetcher``.
    Returns one dict per epoch: lr, loss_avg, auc_avg, impressions."""
    from . import evaluation, ops
    history = []
    for epoch in range(epochs):
        model.train()                                                  # :56
        lr = optimizer.param_groups[0]["lr"]
        loss_sum = torch.zeros((), dtype=torch.float64, device=device)
        auc_sum = torch.zeros((), dtype=torch.float64, device=device)
        bad_rows = torch.zeros((), dtype=torch.int64, device=device)
        seen = 0
        pf = BatchPrefetcher(make_loader(), device)
        for i, (batch, slot) in enumerate(pf):
            loss, out = train_step(model, optimizer, batch)
            auc, _hit = evaluation.row_auc_top1(out, batch["label"])
            pf.release(slot)
            n = out.shape[0]
            loss_sum += loss.double() * n                               # :82 total_loss += loss.item() * B
            auc_sum += auc.double().sum()
            bad_rows += (auc < 0).sum()                                 # one class in a row: roc_auc_score raises (:79)
            seen += n
            if on_batch is not None:
                on_batch(epoch, i, loss, auc)
        if int(bad_rows):                                               # checked once per epoch: no per-batch host sync
            raise ValueError("Only one class present in y_true. ROC AUC score is not defined in that case.")
        # an out-of-range table index / user id: the reference raises IndexError at the offending batch (F.embedding, delta[id]);
        # the kernels clamp and flag, train_step raises at the start of the NEXT step (IndexErrorWatch, no host sync); the last
        # batches of the epoch are covered here, at the epoch's one host sync
        ops.check_index_errors(next(model.parameters()).device)
        rec = {"epoch": epoch, "lr": lr, "impressions": seen,
               "loss_avg": float(loss_sum / max(seen, 1)), "auc_avg": float(auc_sum / max(seen, 1))}
        history.append(rec)
        if ckpt_path is not None:
            evaluation.save_checkpoint(model, ckpt_path.format(epoch=epoch))
    return history
