"""The hot-path operators: ``torch.library`` custom ops (namespace ``nrm``) over the C-ABI kernels of
include/nrm_hotpath.h, plus the differentiable Python entry points the Modules call.

Layering: ``modules.py`` -> functions below -> ``torch.ops.nrm.*`` (schema, fake-tensor shape function,
autograd formula registered with ``torch.library``) -> ``native.call`` (ctypes) -> ``libnrm_hotpath.so``.  Every op has a
CUDA (= ROCm) implementation only: inputs must live on an MI355X, there is no CPU kernel and no eager fallback.

``torch.ops.nrm.pwattn_fwd`` replaces the body of the reference ``PointwiseAttentionExpanded.forward``
(models/attention_model.py:52-97); ``mlp_gelu_fwd`` the reference ``MLP.forward`` (:29-32); ``batch_norm_fwd``,
``softmax_bce_loss`` the head and loss of models/user_model.py:31-43; ``frontend_fwd`` the slicing/embedding front end of
models/user_invariant_interest_model.py:50-79; ``weighted_pool_fwd`` its pool (:86-87).
"""
from __future__ import annotations

import torch

from . import native

EPI_BIAS, EPI_GELU, EPI_DGELU, EPI_MUL = 0, 1, 2, 3
MMA_F32, MMA_BF16, MMA_BF16X3 = 0, 1, 2     # NRM_MMA_* of include/nrm_hotpath.h: arithmetic of the attention's bilinear contraction
DZ_F32, DZ_HL4 = 0, 1                       # NRM_DZ_*: layout in which the dz pass leaves dz (fp32 | bf16 hi/lo pairs)
_MMA_NAMES = {"f32": MMA_F32, "fp32": MMA_F32, "float32": MMA_F32, "bf16": MMA_BF16, "bfloat16": MMA_BF16, "bf16x3": MMA_BF16X3}
_default_mma = MMA_F32


def resolve_mma(mma):
    """'f32' | 'bf16' | 'bf16x3' | NRM_MMA_* | None (= the process-wide default set by ``set_attention_arithmetic``)."""
    if mma is None:
        return _default_mma
    if isinstance(mma, str):
        if mma.lower() not in _MMA_NAMES:
            raise ValueError(f"attention arithmetic {mma!r}: expected 'f32', 'bf16' or 'bf16x3'")
        return _MMA_NAMES[mma.lower()]
    if int(mma) not in (MMA_F32, MMA_BF16, MMA_BF16X3):
        raise ValueError(f"attention arithmetic {mma!r}: expected NRM_MMA_F32 (0), NRM_MMA_BF16 (1) or NRM_MMA_BF16X3 (2)")
    return int(mma)


_default_dense_mma = MMA_F32


def set_dense_arithmetic(mma):
    """Process-wide arithmetic of the dense layers (Linear / MLP / BatchNorm gate GEMMs, w1): 'f32' (default: fp32 MFMA, BASELINE
    config 3) or 'bf16x3' / 'bf16' (bf16 matrix cores with fp32 accumulation, bias and GELU -- BASELINE config 2; GEMMs whose
    reduction width does not fit the resident-row kernel stay fp32).  The attention's own side projections follow the
    attention's arithmetic."""
    global _default_dense_mma
    _default_dense_mma = resolve_mma(mma) if mma is not None else MMA_F32


def set_attention_arithmetic(mma):
    """Process-wide default for attentions that do not carry their own ``mma`` attribute: 'f32' (BASELINE config 3,
    the default), 'bf16' (bf16 MFMA operands, fp32 accumulation) or 'bf16x3' (bf16 MFMA with hi/lo split operands:
    fp32-class accuracy -- the arithmetic BASELINE config 2 is run with)."""
    global _default_mma
    _default_mma = resolve_mma(mma)


def _require_gpu(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise RuntimeError(
                "news_recommendation_model_amd: the hot path only runs on an MI355X (ROCm) device; got a "
                f"{t.device} tensor. Move the model and the inputs to 'cuda' (there is no CPU fallback).")


def _f32c(x):
    return x.to(torch.float32).contiguous()


def _degenerate(shape, *deps):
    """fp32 zeros of `shape` for an input with no rows (empty batch, empty history, no candidates): what the
    reference's ATen ops return there.  The result stays connected, with exactly-zero gradients, to every input
    that requires grad, as the reference's graph would be."""
    tensors = [d for d in deps if isinstance(d, torch.Tensor)]
    _require_gpu(*tensors)
    out = torch.zeros(shape, dtype=torch.float32, device=tensors[0].device)
    for d in tensors:
        if d.requires_grad:
            out = out + d.reshape(-1)[:0].sum().to(torch.float32)     # sum of nothing: 0, but a graph edge
    return out


def _pad4(n):
    return (n + 3) // 4 * 4


def _rows(x):
    """A 2-D fp32 view the GEMM kernels can stream: unit column stride, row stride a multiple of 4 floats,
    16-byte aligned rows.  Anything else is copied once into a zero-padded buffer (padding must be finite)."""
    if (x.dtype == torch.float32 and x.dim() == 2 and x.stride(1) == 1 and x.stride(0) % 4 == 0
            and x.stride(0) >= x.shape[1] and x.data_ptr() % 16 == 0):
        return x
    buf = torch.zeros(x.shape[0], _pad4(x.shape[1]), dtype=torch.float32, device=x.device)
    buf[:, :x.shape[1]] = x
    return buf[:, :x.shape[1]]


def _padded_empty(like, rows, cols):
    """[rows, cols] fp32 view of a fresh [rows, pad4(cols)] buffer (the layout every kernel output uses)."""
    return torch.empty(rows, _pad4(cols), dtype=torch.float32, device=like.device)[:, :cols]


# ------------------------------------------------------------------------------------------------ GEMM plumbing
class _PackedWeights:
    """The GEMM kernels stream their weight operand from a packed image (nrm_gemm_pack).  Weights change once per optimizer
    step, so the images of PERSISTENT weights (model parameters) are kept here, keyed by source pointer + layout, and

    * re-packed lazily, one launch per image, whenever the parameter's autograd version counter has moved
      (``torch.optim`` steps, ``load_state_dict``) -- what every call did before;
    * re-packed ALL TOGETHER by one launch (``nrm_gemm_pack_multi``) when ``repack_persistent()`` is called -- which
      ``trainer.FlatAdam.step()`` does right after its Adam launch (it updates the weights through a raw pointer, so it also
      bumps ``epoch`` to invalidate whatever is not refreshed);
    * simply re-used while nothing changed (inference).
    Temporaries (``owner=None``) are packed per call into a fresh buffer and never cached."""

    def __init__(self):
        self.entries = {}
        self.epoch = 0
        self.generation = 0          # bumped when images are FREED (invalidate_packed_weights): captured graphs that read them are stale


class _PackEntry:
    __slots__ = ("buf", "owner", "owner_ptr", "version", "epoch", "spec")


_packs = _PackedWeights()


def _pack_desc(src, rs, cs, nrows, ncols, src2, sign2, buf, mma=MMA_F32):
    return native.PackDesc(src.data_ptr(), src2.data_ptr() if src2 is not None else None, float(sign2), int(rs), int(cs),
                           int(nrows), int(ncols), buf.data_ptr(), int(mma))


def _pack_desc_of(ent, owner):
    """Descriptor of a cached entry, addresses taken from its live owner (byte offsets were stored, not views)."""
    off, rs, cs, nrows, ncols, off2, sign2, mma = ent.spec
    base = owner.data_ptr()
    return native.PackDesc(base + off, (base + off2) if off2 is not None else None, float(sign2), int(rs), int(cs),
                           int(nrows), int(ncols), ent.buf.data_ptr(), int(mma))


def _pack(src, rs, cs, nrows, ncols, src2=None, sign2=0.0, owner=None, mma=MMA_F32):
    """Packed image of the logical [nrows x ncols] matrix src[r*rs + c*cs] (+ sign2 * src2[same]) for the given arithmetic (fp32
    streaming layout, or bf16 hi [+ lo] MFMA fragments); see _PackedWeights."""
    import weakref
    lib = native.load()
    if owner is None:
        buf = torch.empty(lib.nrm_gemm_packed_floats(nrows, ncols, mma), dtype=torch.float32, device=src.device)
        d = _pack_desc(src, rs, cs, nrows, ncols, src2, sign2, buf, mma)
        native.call("nrm_gemm_pack_multi", _ctypes_ref(d), 1, native.stream_ptr())
        return buf
    key = (src.data_ptr(), int(rs), int(cs), int(nrows), int(ncols), src2.data_ptr() if src2 is not None else 0, float(sign2), int(mma))
    ent = _packs.entries.get(key)
    if ent is not None and (ent.owner() is not owner or ent.owner_ptr != owner.data_ptr()):
        ent = None                                                   # the address now belongs to another tensor
    if ent is not None and ent.version == owner._version and ent.epoch == _packs.epoch:
        return ent.buf
    if ent is None:
        ent = _PackEntry()
        ent.buf = torch.empty(lib.nrm_gemm_packed_floats(nrows, ncols, mma), dtype=torch.float32, device=src.device)
        ent.owner, ent.owner_ptr = weakref.ref(owner), owner.data_ptr()
        # the views are rebuilt from the live owner when needed (a stored view would keep a dead model's storage alive)
        ent.spec = (src.data_ptr() - owner.data_ptr(), rs, cs, nrows, ncols,
                    (src2.data_ptr() - owner.data_ptr()) if src2 is not None else None, sign2, mma)
        _packs.entries[key] = ent
        weakref.finalize(owner, _packs.entries.pop, key, None)     # the entry goes when its parameter goes
    d = _pack_desc(src, rs, cs, nrows, ncols, src2, sign2, ent.buf, mma)
    native.call("nrm_gemm_pack_multi", _ctypes_ref(d), 1, native.stream_ptr())
    ent.version, ent.epoch = owner._version, _packs.epoch
    return ent.buf


def _ctypes_ref(desc_or_array):
    import ctypes
    return ctypes.cast(ctypes.pointer(desc_or_array), ctypes.c_void_p)


def repack_persistent(owners=None):
    """The cached weight images of ``owners`` (an iterable of parameters; None = every live entry) refreshed from their (just
    updated) parameters by ONE launch; entries whose parameter is gone or has been re-seated are dropped.  Called by
    trainer.FlatAdam.step() with ITS parameters: a captured step (trainer.GraphedTrainStep) then only bakes in the addresses of
    the model it was captured for, never those of another live model's entries.  Every other entry is invalidated (``epoch``)
    and re-packed lazily on its next use."""
    _packs.epoch += 1
    mine = None if owners is None else {id(o) for o in owners}
    live = []
    for key, ent in list(_packs.entries.items()):
        owner = ent.owner()
        if owner is None or owner.data_ptr() != ent.owner_ptr:
            _packs.entries.pop(key, None)
            continue
        if mine is not None and id(owner) not in mine:
            continue
        live.append((ent, owner))
    if not live:
        return 0
    arr = (native.PackDesc * len(live))(*[_pack_desc_of(ent, owner) for ent, owner in live])
    native.call("nrm_gemm_pack_multi", _ctypes_ref(arr), len(live), native.stream_ptr())
    for ent, owner in live:
        ent.version, ent.epoch = owner._version, _packs.epoch
    return len(live)


def invalidate_packed_weights():
    """Forget every cached packed weight image.  Needed only after writing to a parameter in a way autograd's version counter
    does not see (``param.data.copy_(...)``, a raw-pointer update): in-place ops on the parameter itself, ``load_state_dict``,
    ``.to()`` and ``torch.optim`` steps are noticed on their own, ``trainer.FlatAdam`` refreshes the images itself."""
    _packs.entries.clear()
    _packs.epoch += 1
    _packs.generation += 1


def packed_weights_generation():
    """Changes whenever cached weight images were released (their memory may be reused): what evaluation.GraphedPredict
    compares before replaying a graph that reads them."""
    return _packs.generation


# Matrix-FLOP meter (bench.py's whole-step roofline): while enabled, every GEMM / contraction launch adds its ALGORITHMIC FLOPs
# (2 M N K; no tile padding, no bf16x3 issue factor) under "dense" or "contraction".
_flop_meter = None


def flop_meter_start():
    global _flop_meter
    _flop_meter = {"dense": 0.0, "contraction": 0.0, "dense_launches": 0, "contraction_launches": 0}


def flop_meter_stop():
    global _flop_meter
    m, _flop_meter = _flop_meter, None
    return m


def _count_flops(kind, flops):
    if _flop_meter is not None:
        _flop_meter[kind] += float(flops)
        _flop_meter[kind + "_launches"] += 1


import os as _os
_FORCE_F32_GEMM = _os.environ.get("NRM_GEMM_F32") == "1"        # diagnostics: every dense / side GEMM on the fp32 kernels


def _dense_mma(M, K, mma):
    """The arithmetic a dense GEMM with M rows and reduction width K really runs in: ``mma`` (None = the process default of
    ``set_dense_arithmetic``) if the bf16 resident-row form takes that K, else fp32."""
    mma = _default_dense_mma if mma is None else mma
    if _FORCE_F32_GEMM or (mma != MMA_F32 and not native.load().nrm_gemm_nt_bf16_supported(int(M), int(K), int(mma))):
        return MMA_F32
    return mma


def _gemm_nt(x, w_src, row_stride, col_stride, n_out, k_red, bias, epilogue, z=None, m=None, src2=None, sign2=0.0, owner=None,
             mma=None):
    """y[M, n_out] = epilogue(x[M, k_red] * Wlogical^T), Wlogical[r, c] = w_src[r*row_stride + c*col_stride]
    (+ sign2 * src2[same]); ``owner`` = the parameter w_src is a view of (its packed image is then cached); ``mma``: arithmetic
    of the products (None = the dense default)."""
    st = native.stream_ptr()
    M = x.shape[0]
    mma = _dense_mma(M, k_red, mma)
    packed = _pack(w_src, row_stride, col_stride, n_out, k_red, src2, sign2, owner, mma)
    ldy = _pad4(n_out)
    y = torch.empty(M, ldy, dtype=torch.float32, device=x.device)
    if epilogue in (EPI_GELU, EPI_MUL):
        z = torch.empty(M, ldy, dtype=torch.float32, device=x.device)
    _count_flops("dense", 2.0 * M * n_out * k_red)
    native.call("nrm_gemm_nt", native.ptr(x), x.stride(0), M, native.ptr(packed), n_out, k_red,
                native.ptr(bias) if bias is not None else None, native.ptr(y), ldy,
                native.ptr(z) if z is not None else None, z.stride(0) if z is not None else 0,
                native.ptr(m) if m is not None else None, m.stride(0) if m is not None else 0, epilogue, mma, st)
    return y[:, :n_out], (z[:, :n_out] if epilogue in (EPI_GELU, EPI_MUL) else None)


def _gemm_tn_slabs(a, b, want_colsum, zero_out=None, mma=None):
    """Partial slabs of (sum_r a[r,i] b[r,j]): ws[s][j][ldws] (TRANSPOSED) and the per-split column sums of a; ``zero_out``
    (the buffer the slab reduction will add to) is zeroed by the same launch.  ``mma``: arithmetic (None = the dense default)."""
    lib = native.load()
    R, ni = a.shape
    nj = b.shape[1]
    mma = _default_dense_mma if mma is None else mma
    if _FORCE_F32_GEMM or (mma != MMA_F32 and (a.stride(0) % 4 or b.stride(0) % 4 or a.data_ptr() % 16 or b.data_ptr() % 16)):
        mma = MMA_F32                                   # the bf16 forms read 16-byte row segments
    nsplit = lib.nrm_gemm_tn_nsplit(ni, nj, R, mma)
    ldws = _pad4(ni)
    ws = torch.empty(nsplit, nj, ldws, dtype=torch.float32, device=a.device)
    cs = torch.empty(nsplit, ldws, dtype=torch.float32, device=a.device) if want_colsum else None
    _count_flops("dense", 2.0 * R * ni * nj)
    native.call("nrm_gemm_tn", native.ptr(a), a.stride(0), ni, native.ptr(b), b.stride(0), nj, R,
                native.ptr(ws), ldws, native.ptr(cs) if cs is not None else None,
                native.ptr(zero_out) if zero_out is not None else None, zero_out.numel() if zero_out is not None else 0,
                mma, native.stream_ptr())
    return ws, cs, nsplit, ldws


# Deferred slab reductions.  Inside ``with deferred_slab_reductions():`` a slab reduction is only recorded and ALL recorded
# ones run as ONE nrm_slab_reduce_multi launch at flush_slab_reductions() -- the gradients they produce are not valid before
# that.  trainer.train_step wraps backward() in it and FlatAdam.collect_grads flushes: nothing reads a weight gradient in
# between, and 14 latency-bound launches per step become one.  Outside the context (any other caller of backward()) every
# reduction runs at once, as before.
# A record keeps the slabs alive and, of the DESTINATION tensors, only their storages: autograd's AccumulateGrad adopts a
# gradient tensor without a copy only while nobody else references the tensor (a reference to it here made AccumulateGrad
# clone the not-yet-reduced buffer); a reference to the storage keeps the memory valid without counting as one.
_deferred = {"on": False, "pending": []}


class deferred_slab_reductions:
    def __enter__(self):
        self.prev, _deferred["on"] = _deferred["on"], True
        return self

    def __exit__(self, *exc):
        _deferred["on"] = self.prev
        if exc[0] is not None:                       # a failed backward: drop what was recorded, its outputs are abandoned
            _deferred["pending"].clear()
        return False


def verify_deferred_targets(params):
    """Every recorded reduction must still point INTO the gradient tensor autograd left on a parameter.  AccumulateGrad adopts
    the buffer an op returned only while it is the sole owner and the parameter had no gradient; a weight used twice in the
    graph (autograd sums two not-yet-reduced buffers into a new tensor), a pre-existing ``p.grad`` (accumulated into early), a
    gradient hook or ``create_graph`` all leave ``p.grad`` in OTHER memory, which the deferred launch would never reach --
    silently wrong gradients.  Raises instead (and drops the records); such callers must run backward() outside
    ``deferred_slab_reductions()`` (``trainer.train_step(..., defer_reductions=False)``)."""
    pend = _deferred["pending"]
    if not pend:
        return
    by_ptr = {p.data_ptr(): p for p in params}
    for rec in pend:
        p = by_ptr.get(rec["target"])
        if p is None or not p.requires_grad:            # a frozen weight (its gradient was dropped by autograd) or an unknown one
            continue
        if p.grad is None or p.grad.untyped_storage().data_ptr() != rec["keep"][0].data_ptr():
            _deferred["pending"] = []
            _join_wgrad_stream()
            raise RuntimeError(
                "deferred slab reductions: a weight gradient was copied or accumulated before its reduction ran (a weight "
                "used more than once in the graph, a gradient that existed before backward(), a gradient hook, or "
                "create_graph): its values would be wrong. Run this backward outside ops.deferred_slab_reductions() -- "
                "trainer.train_step(..., defer_reductions=False).")


def flush_slab_reductions():
    """Run every recorded slab reduction (one launch on the current stream); no-op when nothing is pending."""
    pend, _deferred["pending"] = _deferred["pending"], []
    _join_wgrad_stream()                               # slabs written on the weight-gradient stream
    if not pend:
        return
    descs = (native.SlabDesc * len(pend))()
    for d, rec in zip(descs, pend):
        d.ws, d.nsplit, d.nj, d.ldws, d.ni = rec["ws"].data_ptr(), rec["nsplit"], rec["nj"], rec["ldws"], rec["ni"]
        d.out, d.out_istride, d.out_jstride = rec["out"], rec["out_is"], rec["out_js"]
        d.out2, d.out2_istride, d.out2_jstride, d.sign2 = rec["out2"], rec["out2_is"], rec["out2_js"], rec["sign2"]
        if rec["vec"] is not None:
            d.vec, d.vec_out = rec["vec"].data_ptr(), rec["vec_out"]
    native.call("nrm_slab_reduce_multi", descs, len(pend), native.stream_ptr())


def _slab_reduce(ws, nsplit, nj, ldws, ni, out, out_is, out_js, out2=None, out2_is=0, out2_js=0, sign2=0.0,
                 vec=None, vec_out=None, target=None):
    """out (+)= sum over the split slabs (float atomics: ``out`` / ``out2`` must be zero-initialised).  ``target``: the weight
    whose gradient ``out`` is (verify_deferred_targets checks that autograd really left ``out`` on it)."""
    if _deferred["on"]:
        # (slabs recorded on the second attention's stream need no allocator bookkeeping: that stream waits for the main
        # one before it is given new work, i.e. after the flush that read them)
        _deferred["pending"].append(dict(
            ws=ws, nsplit=nsplit, nj=nj, ldws=ldws, ni=ni, out=out.data_ptr(), out_is=out_is, out_js=out_js,
            out2=out2.data_ptr() if out2 is not None else None, out2_is=out2_is, out2_js=out2_js, sign2=float(sign2),
            vec=vec, vec_out=vec_out.data_ptr() if vec_out is not None else None,
            target=target.data_ptr() if target is not None else None,
            keep=(out.untyped_storage(), out2.untyped_storage() if out2 is not None else None,
                  vec_out.untyped_storage() if vec_out is not None else None)))
        return
    native.call("nrm_slab_reduce", native.ptr(ws), nsplit, nj, ldws, ni, native.ptr(out), out_is, out_js,
                native.ptr(out2) if out2 is not None else None, out2_is, out2_js, float(sign2),
                native.ptr(vec) if vec is not None else None, native.ptr(vec_out) if vec_out is not None else None,
                native.stream_ptr())


# Weight-gradient stream.  dW = dY^T X feeds nothing but the optimizer, so inside ``deferred_slab_reductions()`` (i.e. under
# trainer.train_step with FlatAdam, where nobody reads a weight gradient before collect_grads) the weight-gradient GEMMs can be
# issued on a side stream; flush_slab_reductions() joins it before the reductions read the slabs.  Round 3 measured that as worth
# 0.1 ms at C3 and made it the default from 20 000 rows; with the round-4/5 kernels it is a LOSS where it applies: C3 eager 31.04 /
# 31.04 / 31.18 ms without it against 31.63 / 31.93 / 31.76 with it (same box, alternating; after 40 warm-up steps 31.69 against
# 32.11), the captured step the same either way (31.64 / 31.47), C5 116.4 / 116.5 (scripts/_diag/r5_eager_streams*.sh).  Two big
# GEMMs sharing the chip take the sum of their times, and the cross-stream event waits are not free.  So it is OFF by default;
# NRM_WGRAD_STREAM=1 turns it on (tests keep the path covered).
WGRAD_STREAM_MIN_ROWS = 0
_wgrad = {"streams": {}, "used": set()}


class _wgrad_stream:
    """with _wgrad_stream(a, b) as on_side: ... -- runs the block on the weight-gradient stream when that applies."""

    def __init__(self, *inputs):
        import os
        forced = os.environ.get("NRM_WGRAD_STREAM")
        self.inputs = inputs
        self.on = _deferred["on"] and forced == "1" and inputs[0].shape[0] >= WGRAD_STREAM_MIN_ROWS

    def __enter__(self):
        if not self.on:
            return False
        dev = self.inputs[0].device
        key = dev.index if dev.index is not None else torch.cuda.current_device()
        side = _wgrad["streams"].get(key)
        if side is None:
            side = _wgrad["streams"][key] = torch.cuda.Stream(device=dev)
        self.main = torch.cuda.current_stream(dev)
        side.wait_stream(self.main)                      # the operands exist once the main stream gets here
        for t in self.inputs:
            t.record_stream(side)                        # allocated on the main stream, read by side-stream kernels
        _wgrad["used"].add(key)
        self.ctx = torch.cuda.stream(side)
        self.ctx.__enter__()
        return True

    def __exit__(self, *exc):
        if self.on:
            self.ctx.__exit__(*exc)
        return False


def _join_wgrad_stream():
    """The current stream waits for every weight-gradient stream that was given work since the last join."""
    for key in list(_wgrad["used"]):
        torch.cuda.current_stream().wait_stream(_wgrad["streams"][key])
    _wgrad["used"].clear()


def _gemm_tn(a, b, want_colsum, target=None, mma=None):
    """(sum_r a[r,i] b[r,j]) as a contiguous [ni, nj], and optionally sum_r a[r,i]: the split-M GEMM plus ONE
    reduce+transpose launch that writes the gradient in place (no ATen sum/t/contiguous)."""
    ni, nj = a.shape[1], b.shape[1]
    ctx = _wgrad_stream(a, b)
    with ctx as side:
        c = torch.empty(ni, nj, dtype=torch.float32, device=a.device)      # zeroed by the GEMM launch itself
        ws, cs, nsplit, ldws = _gemm_tn_slabs(a, b, want_colsum, zero_out=c, mma=mma)
        colsum = torch.empty(ni, dtype=torch.float32, device=a.device) if want_colsum else None
        _slab_reduce(ws, nsplit, nj, ldws, ni, c, nj, 1, vec=cs, vec_out=colsum, target=target)
        if side:                                           # read (reduced, gathered) and released on the main stream after the join
            for t in (c, ws, cs, colsum):
                if t is not None:
                    t.record_stream(ctx.main)
    return c, colsum


# ------------------------------------------------------------------------------------------------ op registry
_LIB = torch.library.Library("nrm", "DEF")


def _op(name, schema, impl, fake):
    _LIB.define(name + schema)
    _LIB.impl(name, impl, "CUDA")
    torch.library.register_fake("nrm::" + name, fake, lib=_LIB)
    return getattr(torch.ops.nrm, name)


# ------------------------------------------------------------------------------------------------ pointwise attention
def _pwattn_fwd_impl(t, h, w1, b1, w2, b2, save_z, mma):
    """s[b,t,h] = fc2(GELU(fc1(cat[h, t, t-h, t*h]))) with fc1 = [W_h|W_t|W_d|W_p] re-associated as
    z = h(W_h-W_d)^T + b1 + t(W_t+W_d)^T + sum_d W_p[:,d] t_d h_d  (SURVEY.md §8 a7)."""
    _require_gpu(t, h, w1, b1, w2, b2)
    if save_z:
        _chain["end"] = None               # a training forward: whatever an earlier backward left there is stale (see _chain below)
    B, T, D = t.shape
    H = h.shape[1]
    if h.shape[0] != B or h.shape[2] != D or tuple(w1.shape) != (D, 4 * D) or D % 4:
        raise RuntimeError(f"pointwise attention: target {tuple(t.shape)}, history {tuple(h.shape)}, "
                           f"fc1 {tuple(w1.shape)} do not agree (feature width must be a multiple of 4)")
    w1_arg = w1
    t, h, w1, b1 = _f32c(t), _f32c(h), _f32c(w1), _f32c(b1)
    w2v, b2 = _f32c(w2).reshape(-1), _f32c(b2).reshape(-1)
    # side projections u = h (W_h - W_d)^T + b1, v = t (W_t + W_d)^T: the difference / sum is formed by the pack kernel
    own = w1 if w1 is w1_arg else None
    w_h, w_t, w_d = w1[:, :D], w1[:, D:2 * D], w1[:, 2 * D:3 * D]
    u, _ = _gemm_nt(h.reshape(B * H, D), w_h, 4 * D, 1, D, D, b1, EPI_BIAS, src2=w_d, sign2=-1.0, owner=own, mma=mma)     # [B*H, D]
    v, _ = _gemm_nt(t.reshape(B * T, D), w_t, 4 * D, 1, D, D, None, EPI_BIAS, src2=w_d, sign2=1.0, owner=own, mma=mma)   # [B*T, D]
    st = native.stream_ptr()
    packed = torch.empty(native.load().nrm_pwattn_packed_floats(D), dtype=torch.float32, device=t.device)
    native.call("nrm_pwattn_pack_wp", native.ptr(w1), 4 * D, D, mma, native.ptr(packed), st)
    s = torch.empty(B, T, H, dtype=torch.float32, device=t.device)
    z = torch.empty((B, T, H, D) if save_z else (0,), dtype=torch.float32, device=t.device)
    _count_flops("contraction", 2.0 * B * T * H * D * D)
    native.call("nrm_pwattn_fwd", native.ptr(t), native.ptr(h), native.ptr(u), native.ptr(v),
                native.ptr(packed), native.ptr(w2v), native.ptr(b2),
                native.ptr(z) if save_z else None, native.ptr(s), B, T, H, D, mma, st)
    return s, z


def _pwattn_fwd_fake(t, h, w1, b1, w2, b2, save_z, mma):
    B, T, D = t.shape
    H = h.shape[1]
    return (t.new_empty((B, T, H), dtype=torch.float32),
            t.new_empty((B, T, H, D) if save_z else (0,), dtype=torch.float32))


pwattn_fwd = _op("pwattn_fwd", "(Tensor t, Tensor h, Tensor fc1_weight, Tensor fc1_bias, Tensor fc2_weight, Tensor fc2_bias, "
                 "bool save_z, int mma) -> (Tensor, Tensor)", _pwattn_fwd_impl, _pwattn_fwd_fake)


def _pwattn_bwd_impl(ds, t, h, w1, w2, z, mma, need_dt, need_dh):
    """All gradients of one attention from ds [B,T,H] and the saved pre-activation z [B,T,H,D], which is overwritten
    in place by dz (schema: Tensor(a!))."""
    return _attn_bwd_core(ds, t, h, w1, w2, z, mma, need_dt, need_dh)


# Order of the two attentions' backward contractions when they run on two streams (round 5).  The label attention's chain
# (dz -> (b,t) pass -> (b,h) pass) is followed on ITS stream by ~1.3 ms of atomic- / latency-bound work that depends on it (w1's
# backward, the front-end scatter, the category sort); the text+image attention's dW_p-only pass depends on nothing but its own dz
# and feeds nothing but the optimizer.  Issued as soon as its dz is ready it shares the matrix pipe with the (b,t) pass -- two
# MFMA-bound kernels take the sum of their times -- and the tail then runs alone.  So a dW_p-only contraction waits for the end of a
# full chain that was enqueued on ANOTHER stream earlier in the same step: it then overlaps the tail instead (C3 eager: DESIGN.md
# section 4f).  A chain's end is only ever waited for inside the backward pass it belongs to: every training forward clears it (and
# trainer.train_step opens a step, begin_step); NRM_DW_LAST=0 switches the wait off.
_chain = {"token": 0, "end": None}


def begin_step():
    _chain["token"] += 1
    _chain["end"] = None


def _note_chain_end():
    ev = torch.cuda.Event()
    ev.record()
    _chain["end"] = (_chain["token"], ev, torch.cuda.current_stream(), torch.cuda.is_current_stream_capturing())


# Measured (same box, alternating, eager): C3 (614 M z elements) 30.98 / 31.06 / 31.14 ms with the wait against 31.24 / 31.32 / 31.10
# without, 31.47 against 31.74 after 40 warm-up steps, captured step 31.33 against 31.47; C5 (1.6 G) 117.9 against 116.6 -- its tail is
# 2 % of a contraction, there is nothing to hide; C2 and smaller: their kernels do not fill the chip, side by side is the gain there.
DW_LAST_MIN_ELEMS, DW_LAST_MAX_ELEMS = 200_000_000, 1_000_000_000


def _wait_for_foreign_chain(z_elems):
    import os
    end = _chain["end"]
    forced = os.environ.get("NRM_DW_LAST")
    if end is None or forced == "0" or (forced != "1" and not (DW_LAST_MIN_ELEMS <= z_elems <= DW_LAST_MAX_ELEMS)):
        return
    token, ev, stream, capturing = end
    cur = torch.cuda.current_stream()
    if token == _chain["token"] and stream != cur and capturing == torch.cuda.is_current_stream_capturing():
        cur.wait_event(ev)


# fp32 backward of the bilinear term: the dP walk (one contraction for dt AND dh + the dW_p-only pass) instead of the two E-form
# passes.  NRM_BWD_DP=1 forces it wherever the library has it (D % 4 == 0, H >= 16), =0 never; default: by size (DP_MIN_ELEMS z elements).
# Measured per attention (ms, dP + dW_p-only against (b,t) + (b,h), same box): C3 3.98 + 4.06 against 4.38 + 4.04; C5 19.0 + 19.9 against
# 20.5 + 20.1; C2's shape in fp32 0.60 + 0.63 against 0.72 + 0.62; the reference's default sizes (49 M elements) 0.085 + 0.073 against
# 0.080 + 0.079 -- equal, and one launch (the W_p^T pack) more: the E-form stays there.
DP_MIN_ELEMS = 100_000_000


def _use_dp_walk(lib, B, T, H, D, mma):
    import os
    forced = os.environ.get("NRM_BWD_DP")
    if mma != MMA_F32 or forced == "0" or not lib.nrm_pwattn_bwd_dp_supported(D, H):
        return False
    return forced == "1" or B * T * H * D >= DP_MIN_ELEMS


def _attn_bwd_core(ds, t, h, w1, w2, z, mma, need_dt=True, need_dh=True, acc=None):
    """``need_dt`` / ``need_dh``: whether the target / history rows want a gradient.  The text+image attention of the model reads
    raw input columns (reference user_invariant_interest_model.py:63-64,78: no parameter upstream), so autograd asks for neither
    -- as in the reference, where those products are never formed -- and the (b,h)-grouped contraction pass, the two
    side-projection GEMMs and the dt epilogue are skipped: the weight gradients only need dz, du, dv and the dW_p pass.
    Skipped gradients are returned as empty tensors.
    ``acc``: [D + 4] floats (dw2 | db2 | pad) that an earlier launch on this stream has ALREADY zeroed (the pool's rowdot
    kernel in the merged pool + attention backward); None: allocated and zeroed here."""
    _require_gpu(ds, t, h, w1, w2, z)
    B, T, D = t.shape
    H = h.shape[1]
    st = native.stream_ptr()
    w1_arg = w1
    t, h, w1, ds = _f32c(t), _f32c(h), _f32c(w1), _f32c(ds)
    own = w1 if w1 is w1_arg else None
    w2v = _f32c(w2).reshape(-1)
    dev = t.device
    if acc is None:
        acc = torch.zeros(D + 4, dtype=torch.float32, device=dev)
    dw2, db2 = acc[:D], acc[D:D + 1]                   # the dz pass accumulates both (db2 = sum ds); returned as ONE tensor
    # one pass over z: z -> dz in place, du = sum_t dz, dv = sum_h dz, dw2.  The bf16 arithmetics with a resident-W backward
    # (D <= 256) get dz as bf16 hi/lo pairs (NRM_DZ_HL4): the contraction kernels then read MFMA-ready operands
    lib = native.load()
    rw = mma != MMA_F32 and bool(lib.nrm_pwattn_bwd_rw_supported(D, mma))
    du = torch.empty(B, H, D, dtype=torch.float32, device=dev)
    dv = torch.empty(B, T, D, dtype=torch.float32, device=dev)
    native.call("nrm_pwattn_bwd_dz", native.ptr(z), native.ptr(ds), native.ptr(w2v), native.ptr(dw2), native.ptr(db2),
                native.ptr(du), native.ptr(dv), B, T, H, D, DZ_HL4 if rw else DZ_F32, st)
    dz = z
    w_h, w_t, w_d = w1[:, :D], w1[:, D:2 * D], w1[:, 2 * D:3 * D]
    du2, dv2 = du.reshape(B * H, D), dv.reshape(B * T, D)
    # fc1 gradient [D, 4D] = [da_h | da_t | da_t - da_h | dW_p]: every block is written in place by a slab reduction
    dw1 = torch.empty(D, 4 * D, dtype=torch.float32, device=dev)                    # zeroed by the first GEMM launch below
    db1 = torch.empty(D, dtype=torch.float32, device=dev)
    h2, t2 = h.reshape(B * H, D), t.reshape(B * T, D)
    ctx = _wgrad_stream(du2, h2, dv2, t2, dw1)          # (big batches under train_step: on the weight-gradient stream)
    with ctx as side:
        ws, cs, ns, ldws = _gemm_tn_slabs(du2, h2, True, zero_out=dw1, mma=mma)  # du^T h, db1 = column sums of du
        _slab_reduce(ws, ns, D, ldws, D, dw1, 4 * D, 1, out2=dw1[:, 2 * D:], out2_is=4 * D, out2_js=1, sign2=-1.0,
                     vec=cs, vec_out=db1, target=w1_arg)
        ws2, _, ns, ldws = _gemm_tn_slabs(dv2, t2, False, mma=mma)
        _slab_reduce(ws2, ns, D, ldws, D, dw1[:, D:], 4 * D, 1, out2=dw1[:, 2 * D:], out2_is=4 * D, out2_js=1, sign2=1.0, target=w1_arg)
        if side:
            for x in (ws, cs, ws2):
                x.record_stream(ctx.main)
    # du (W_h - W_d), dv (W_t + W_d): the transposed orientation of the same two combinations   (D % 4 == 0: contiguous results)
    none = torch.empty(0, dtype=torch.float32, device=dev)
    dh = (_gemm_nt(du2, w_h, 1, 4 * D, D, D, None, EPI_BIAS, src2=w_d, sign2=-1.0, owner=own, mma=mma)[0].reshape(B, H, D)
          if need_dh else none)
    dt = (_gemm_nt(dv2, w_t, 1, 4 * D, D, D, None, EPI_BIAS, src2=w_d, sign2=1.0, owner=own, mma=mma)[0].reshape(B, T, D)
          if need_dt else none)
    nsplit = lib.nrm_pwattn_bwd_nsplit(B, T, H, D, mma)
    wsp = torch.empty(nsplit, D, D, dtype=torch.float32, device=dev)
    wp = w1[:, 3 * D:]                                   # view, row stride 4D
    if rw:
        # resident-W form: dt and dh from ONE contraction dP = dz W_p (dz read once), then the (b,t)-grouped pass without its dt
        # epilogue for the dW_p slabs
        if need_dt or need_dh:
            img = torch.empty(lib.nrm_pwattn_bwd_rw_packed_floats(D, mma), dtype=torch.float32, device=dev)
            native.call("nrm_pwattn_bwd_rw_pack", native.ptr(w1), 4 * D, D, mma, native.ptr(img), st)
            # (one of the two unwanted: the kernel forms both anyway; the unwanted one goes to a scratch buffer)
            dt_ = dt if need_dt else torch.empty(B, T, D, dtype=torch.float32, device=dev)
            dh_ = dh if need_dh else torch.empty(B, H, D, dtype=torch.float32, device=dev)
            _count_flops("contraction", 2.0 * B * T * H * D * D)
            native.call("nrm_pwattn_bwd_rw_dtdh", native.ptr(dz), native.ptr(t), native.ptr(h), native.ptr(img), native.ptr(dt_),
                        native.ptr(dh_), B, T, H, D, mma, st, tag="pwattn_bwd_rw_dtdh")
        _count_flops("contraction", 2.0 * B * T * H * D * D)
        if not (need_dt or need_dh):
            _wait_for_foreign_chain(B * T * H * D)
        native.call("nrm_pwattn_bwd_contract", native.ptr(dz), native.ptr(t), native.ptr(h), native.ptr(wp), 4 * D, None, None,
                    native.ptr(wsp), B, T, H, D, 4, mma, DZ_HL4, st, tag="pwattn_bwd_e_bt")
        if need_dt or need_dh:
            _note_chain_end()
    elif need_dt and need_dh and _use_dp_walk(lib, B, T, H, D, mma):
        # fp32, both row gradients wanted: dt and dh from ONE contraction dP = dz W_p (csrc/pwattn_bwd_dp.hip: the forward's
        # streaming skeleton walking the candidates), then the (b,t)-grouped pass without its dt epilogue for the dW_p slabs
        img = torch.empty(lib.nrm_pwattn_bwd_dp_packed_floats(D, H), dtype=torch.float32, device=dev)
        native.call("nrm_pwattn_bwd_dp_pack", native.ptr(w1), 4 * D, D, H, native.ptr(img), st)
        _count_flops("contraction", 2.0 * B * T * H * D * D)
        native.call("nrm_pwattn_bwd_dp_dtdh", native.ptr(dz), native.ptr(t), native.ptr(h), native.ptr(img), native.ptr(dt),
                    native.ptr(dh), B, T, H, D, st, tag="pwattn_bwd_dp_dtdh")
        _count_flops("contraction", 2.0 * B * T * H * D * D)
        native.call("nrm_pwattn_bwd_contract", native.ptr(dz), native.ptr(t), native.ptr(h), native.ptr(wp), 4 * D, None, None,
                    native.ptr(wsp), B, T, H, D, 4, mma, DZ_F32, st, tag="pwattn_bwd_e_dw")
        _note_chain_end()
    else:
        # two launches: (b,t)-grouped -> dt + dW_p slabs, (b,h)-grouped -> dh (issued separately so that
        # bench.py can time each kernel with its own event pair); the second one only when the history wants a gradient
        # (without a target gradient the first one runs without its dt epilogue: passes = 4)
        for passes, tag in ((1, "pwattn_bwd_e_bt") if need_dt else (4, "pwattn_bwd_e_dw"), (2, "pwattn_bwd_e_bh")):
            if passes == 2 and not need_dh:
                continue
            _count_flops("contraction", 2.0 * B * T * H * D * D)
            if not (need_dt or need_dh):
                _wait_for_foreign_chain(B * T * H * D)     # the dW_p-only pass: behind the other attention's chain (see _chain)
            native.call("nrm_pwattn_bwd_contract", native.ptr(dz), native.ptr(t), native.ptr(h),
                        native.ptr(wp), 4 * D, native.ptr(dt) if need_dt else None, native.ptr(dh) if need_dh else None,
                        native.ptr(wsp), B, T, H, D, passes, mma, DZ_F32, st, tag=tag)
        if need_dt or need_dh:
            _note_chain_end()
    _slab_reduce(wsp, nsplit, D, D, D, dw1[:, 3 * D:], 4 * D, 1, target=w1_arg)    # slabs hold dW_p^T: ws[s][d][k] -> dw1[k, 3D + d]
    return dt, dh, dw1, db1, acc


def _pwattn_bwd_fake(ds, t, h, w1, w2, z, mma, need_dt=True, need_dh=True):
    B, T, D = t.shape
    H = h.shape[1]
    f = lambda *shape: t.new_empty(shape, dtype=torch.float32)      # noqa: E731
    return (f(B, T, D) if need_dt else f(0)), (f(B, H, D) if need_dh else f(0)), f(D, 4 * D), f(D), f(D + 4)


# returns (dt, dh, d fc1_weight, d fc1_bias, [d fc2_weight (D) | d fc2_bias (1) | pad (3)])
pwattn_bwd = _op("pwattn_bwd", "(Tensor ds, Tensor t, Tensor h, Tensor fc1_weight, Tensor fc2_weight, Tensor(a!) z, int mma, "
                 "bool need_dt, bool need_dh) -> (Tensor, Tensor, Tensor, Tensor, Tensor)", _pwattn_bwd_impl, _pwattn_bwd_fake)


# The backward overwrites the saved [B,T,H,D] pre-activation in place (2.46 GB at C3: no second copy).  A SECOND backward through the
# same graph (autograd.grad(..., retain_graph=True), which the reference's autograd allows: models/attention_model.py:92) therefore
# finds the buffer spent -- and RECOMPUTES the pre-activation from the saved inputs with one extra forward kernel (round 5; rounds
# 1-4 raised).  The usual single backward pays nothing for this.  set_retain_attention_graph(True) selects the other trade: every
# backward works on a COPY of z (one extra [B,T,H,D] buffer and one copy per backward, no recomputation).
_retain_attention_graph = False


def set_retain_attention_graph(on=True):
    global _retain_attention_graph
    prev, _retain_attention_graph = _retain_attention_graph, bool(on)
    return prev


def _consume_z(ctx, z, t, h, w1, b1, w2):
    """The dz buffer of this backward: z itself on the first walk of the graph; a copy with set_retain_attention_graph; on a later
    walk (the buffer now holds the first walk's dz) the pre-activation recomputed from the saved inputs."""
    if _retain_attention_graph:
        return z.detach().clone()
    if ctx.consumed:
        with torch.no_grad():
            b2 = torch.zeros(1, dtype=torch.float32, device=t.device)            # (does not enter z)
            return _pwattn_fwd_impl(t.detach(), h.detach(), w1.detach(), b1.detach(), w2.detach(), b2, True, ctx.mma)[1]
    ctx.consumed = True
    return z.detach()


def _pwattn_setup(ctx, inputs, output):
    t, h, w1, b1, w2, b2, save_z, mma = inputs
    s, z = output
    ctx.set_materialize_grads(False)
    ctx.save_z = save_z
    ctx.mma = mma
    ctx.consumed = False
    ctx.w2_shape, ctx.b2_shape = tuple(w2.shape), tuple(b2.shape)
    if save_z:
        ctx.save_for_backward(t, h, w1, w2, z, b1)


def _pwattn_backward(ctx, ds, _dz):
    if ds is None:
        return None, None, None, None, None, None, None, None
    if not ctx.save_z:
        raise RuntimeError("pointwise attention: the forward ran with save_z=False (no-grad / inference call); there "
                           "is nothing to differentiate through")
    t, h, w1, w2, z, b1 = ctx.saved_tensors
    need = ctx.needs_input_grad
    dt, dh, dw1, db1, dw2b2 = pwattn_bwd(ds, t, h, w1, w2, _consume_z(ctx, z, t, h, w1, b1, w2), ctx.mma, bool(need[0]), bool(need[1]))
    D = t.shape[2]
    return ((dt if need[0] else None), (dh if need[1] else None), dw1, db1, dw2b2[:D].reshape(ctx.w2_shape),
            dw2b2[D:D + 1].reshape(ctx.b2_shape), None, None)


torch.library.register_autograd("nrm::pwattn_fwd", _pwattn_backward, setup_context=_pwattn_setup, lib=_LIB)


def pointwise_attention_scores(target, history, fc1_weight, fc1_bias, fc2_weight, fc2_bias, mma=None):
    """[B,T,D] x [B,H,D] -> [B,T,H] scores (fp32).  ``mma``: arithmetic of the bilinear contraction ('f32' | 'bf16' |
    None = process default); the side projections, GELU, fc2 and all reductions are fp32 either way.

    The kernels need the feature width to be a multiple of 4 (float4 rows).  Any other D is zero-padded here with
    differentiable ops: padded features contribute exactly 0 to every term (their fc1 rows/columns and fc2 weights
    are 0, gelu(0) = 0), and autograd slices the gradients back."""
    D = target.shape[-1]
    if target.shape[0] * target.shape[1] * history.shape[1] == 0:
        return _degenerate((target.shape[0], target.shape[1], history.shape[1]), target, history, fc1_weight, fc1_bias,
                           fc2_weight, fc2_bias)
    args = (target, history, fc1_weight, fc1_bias, fc2_weight, fc2_bias)
    _require_gpu(*args)
    # the [B,T,H,D] pre-activation is only kept when something will differentiate through the call
    save_z = torch.is_grad_enabled() and any(a.requires_grad for a in args)
    if D % 4:
        P = _pad4(D) - D
        pad = torch.nn.functional.pad
        blocks = [pad(fc1_weight[:, i * D:(i + 1) * D], (0, P, 0, P)) for i in range(4)]     # [D4, D4] each
        args = (pad(target.to(torch.float32), (0, P)), pad(history.to(torch.float32), (0, P)), torch.cat(blocks, dim=1),
                pad(fc1_bias, (0, P)), pad(fc2_weight.reshape(1, D), (0, P)), fc2_bias)
    return pwattn_fwd(*args, save_z, resolve_mma(mma))[0]


# ------------------------------------------------------------------------------------------------ dense layers
def _linear_fwd_impl(x, weight, bias, gelu):
    _require_gpu(x, weight, bias)
    x = _rows(x)
    w = _f32c(weight)
    N, K = w.shape
    if x.shape[1] != K:
        raise RuntimeError(f"linear: input has {x.shape[1]} features, weight expects {K}")
    b = _f32c(bias) if bias is not None else None
    y, z = _gemm_nt(x, w, K, 1, N, K, b, EPI_GELU if gelu else EPI_BIAS, owner=w if w is weight else None)
    return y, (z if gelu else y.new_empty((0,)))


def _linear_fwd_fake(x, weight, bias, gelu):
    M, N = x.shape[0], weight.shape[0]
    y = _padded_empty(x, M, N)
    return y, (_padded_empty(x, M, N) if gelu else x.new_empty((0,), dtype=torch.float32))


linear_fwd = _op("linear_fwd", "(Tensor x, Tensor weight, Tensor? bias, bool gelu) -> (Tensor, Tensor)",
                 _linear_fwd_impl, _linear_fwd_fake)


def _linear_bwd_impl(dy, x, weight, z, has_bias, need_dx, need_dw):
    _require_gpu(dy, x, weight)
    x, w = _rows(x), _f32c(weight)
    N, K = w.shape
    if z.numel():
        dy = torch.ops.aten.gelu_backward(dy, z)            # exact-erf GELU' (elementwise; the fused path is mlp_gelu)
    dy = _rows(dy)
    dev = x.device
    dw = db = dx = None
    if need_dw:
        dw, db = _gemm_tn(dy, x, has_bias, target=weight)   # dW[n,k] = sum_m dy[m,n] x[m,k]
    if need_dx:
        dx, _ = _gemm_nt(dy, w, 1, K, K, N, None, EPI_BIAS, owner=w if w is weight else None)  # dX = dY W : rows of the packed operand = k
    e = lambda: torch.empty((0,), dtype=torch.float32, device=dev)      # noqa: E731
    return (dx if dx is not None else e()), (dw if dw is not None else e()), (db if db is not None else e())


def _linear_bwd_fake(dy, x, weight, z, has_bias, need_dx, need_dw):
    M, (N, K) = x.shape[0], weight.shape
    e = lambda: x.new_empty((0,), dtype=torch.float32)                  # noqa: E731
    return (_padded_empty(x, M, K) if need_dx else e(), x.new_empty((N, K), dtype=torch.float32) if need_dw else e(),
            x.new_empty((N,), dtype=torch.float32) if (need_dw and has_bias) else e())


linear_bwd = _op("linear_bwd", "(Tensor dy, Tensor x, Tensor weight, Tensor z, bool has_bias, bool need_dx, bool need_dw) -> "
                 "(Tensor, Tensor, Tensor)", _linear_bwd_impl, _linear_bwd_fake)


def _linear_setup(ctx, inputs, output):
    x, weight, bias, gelu = inputs
    ctx.set_materialize_grads(False)
    ctx.has_bias = bias is not None
    ctx.save_for_backward(x, weight, output[1])


def _linear_backward(ctx, dy, _dz):
    if dy is None:
        return None, None, None, None
    x, weight, z = ctx.saved_tensors
    need = ctx.needs_input_grad
    need_dw = need[1] or (ctx.has_bias and need[2])
    dx, dw, db = linear_bwd(dy, x, weight, z, ctx.has_bias, need[0], need_dw)
    return (dx if need[0] else None), (dw if need[1] else None), (db if (ctx.has_bias and need[2]) else None), None


torch.library.register_autograd("nrm::linear_fwd", _linear_backward, setup_context=_linear_setup, lib=_LIB)


def linear(x, weight, bias=None, gelu=False):
    """nn.Linear (optionally followed by exact GELU) on the last dimension of x."""
    lead = x.shape[:-1]
    if x.numel() == 0 and x.shape[-1] == weight.shape[1]:
        return _degenerate((*lead, weight.shape[0]), x, weight, bias)
    _require_gpu(x, weight)
    y = linear_fwd(x.reshape(-1, x.shape[-1]), weight, bias, bool(gelu))[0]
    return y.reshape(*lead, weight.shape[0])


# ReLU(Linear) with a handful of columns (the instant-interest layer): dedicated kernels instead of a zero-padded K = 4 GEMM
SMALL_LINEAR_MAX_K, SMALL_LINEAR_MAX_N = 4, 8


def _small_linear_relu_fwd_impl(x, weight, bias):
    _require_gpu(x, weight)
    N, K = weight.shape
    if x.dtype not in (torch.float32, torch.float64):
        x = x.to(torch.float32)
    x = x.contiguous()
    R = x.shape[0]
    w, b = _f32c(weight), (_f32c(bias) if bias is not None else None)
    y = torch.empty(R, _pad4(N), dtype=torch.float32, device=x.device)
    native.call("nrm_small_linear_relu_fwd", native.ptr(x), 1 if x.dtype == torch.float64 else 0, native.ptr(w),
                native.ptr(b) if b is not None else None, native.ptr(y), R, K, N, y.stride(0), native.stream_ptr())
    return y[:, :N]


small_linear_relu_fwd = _op("small_linear_relu_fwd", "(Tensor x, Tensor weight, Tensor? bias) -> Tensor", _small_linear_relu_fwd_impl,
                            lambda x, weight, bias: _padded_empty(x, x.shape[0], weight.shape[0]))


def _small_linear_relu_bwd_impl(dy, x, weight, bias):
    """-> [N*K + N] floats: d weight (row-major) followed by d bias."""
    _require_gpu(dy, x, weight)
    N, K = weight.shape
    if x.dtype not in (torch.float32, torch.float64):
        x = x.to(torch.float32)
    x = x.contiguous()
    w, b = _f32c(weight), (_f32c(bias) if bias is not None else None)
    if not (dy.dtype == torch.float32 and dy.dim() == 2 and dy.stride(1) == 1 and dy.stride(0) >= N):
        dy = _f32c(dy)                                 # (a column block of the head gradient is read in place)
    dwb = torch.zeros(N * K + N, dtype=torch.float32, device=x.device)
    native.call("nrm_small_linear_relu_bwd", native.ptr(x), 1 if x.dtype == torch.float64 else 0, native.ptr(w),
                native.ptr(b) if b is not None else None, native.ptr(dy), dy.stride(0) if dy.shape[0] > 1 else N, x.shape[0], K, N,
                native.ptr(dwb), native.stream_ptr())
    return dwb


small_linear_relu_bwd = _op("small_linear_relu_bwd", "(Tensor dy, Tensor x, Tensor weight, Tensor? bias) -> Tensor", _small_linear_relu_bwd_impl,
                            lambda dy, x, weight, bias: x.new_empty((weight.numel() + weight.shape[0],), dtype=torch.float32))


def _small_linear_setup(ctx, inputs, output):
    ctx.set_materialize_grads(False)
    ctx.has_bias = inputs[2] is not None
    ctx.save_for_backward(*[t for t in inputs if t is not None])


def _small_linear_backward(ctx, dy):
    if dy is None:
        return None, None, None
    x, weight, *rest = ctx.saved_tensors
    if ctx.needs_input_grad[0]:
        raise RuntimeError("small_linear_relu: the input has no gradient path (raw popularity scalars); use ops.linear + relu")
    dwb = small_linear_relu_bwd(dy, x, weight, rest[0] if rest else None)
    n = weight.numel()
    return None, dwb[:n].view(weight.shape), (dwb[n:] if ctx.has_bias else None)


torch.library.register_autograd("nrm::small_linear_relu_fwd", _small_linear_backward, setup_context=_small_linear_setup, lib=_LIB)


def small_linear_relu(x, weight, bias=None):
    """relu(nn.Linear(x)) on the last dimension for K <= 4 inputs, N <= 8 outputs and an input without a gradient; other
    shapes go through ops.linear + relu."""
    N, K = weight.shape
    if (K > SMALL_LINEAR_MAX_K or N > SMALL_LINEAR_MAX_N or x.requires_grad or x.numel() == 0 or x.shape[-1] != K):
        return torch.relu(linear(x if x.dtype == torch.float32 else x.to(torch.float32), weight, bias))
    _require_gpu(x, weight)
    lead = x.shape[:-1]
    y = small_linear_relu_fwd(x.reshape(-1, K), weight, bias)
    return y.reshape(*lead, N) if y.is_contiguous() else y.unflatten(0, tuple(lead))


def _mlp_gelu_fwd_impl(x, w1, b1, w2, b2, mul):
    """fc2(gelu(fc1(x))) [* mul] for a 2-D x (reference models/attention_model.py:29-32 with the default activation):
    two GEMMs, bias + exact GELU fused into the first (its pre-activation saved), bias and -- for the gate of
    user_model.py:33 -- the product with the raw concat fused into the second."""
    _require_gpu(x, w1, w2, mul)
    x = _rows(x)
    o1, o2 = w1, w2
    w1, w2 = _f32c(w1), _f32c(w2)
    o1, o2 = (w1 if w1 is o1 else None), (w2 if w2 is o2 else None)
    N1, K1 = w1.shape
    N2, K2 = w2.shape
    if x.shape[1] != K1 or K2 != N1:
        raise RuntimeError(f"mlp: input has {x.shape[1]} features, fc1 expects {K1}, fc2 expects {K2} hidden")
    hidden, z = _gemm_nt(x, w1, K1, 1, N1, K1, _f32c(b1) if b1 is not None else None, EPI_GELU, owner=o1)
    if mul is None:
        y, _ = _gemm_nt(hidden, w2, K2, 1, N2, K2, _f32c(b2) if b2 is not None else None, EPI_BIAS, owner=o2)
        pre = y.new_empty((0,))
    else:
        m = _rows(mul)
        if tuple(m.shape) != (x.shape[0], N2):
            raise RuntimeError(f"mlp: multiplier {tuple(m.shape)} does not match the output {(x.shape[0], N2)}")
        y, pre = _gemm_nt(hidden, w2, K2, 1, N2, K2, _f32c(b2) if b2 is not None else None, EPI_MUL, m=m, owner=o2)
    return y, hidden, z, pre


def _mlp_gelu_fwd_fake(x, w1, b1, w2, b2, mul):
    M, N1, N2 = x.shape[0], w1.shape[0], w2.shape[0]
    return (_padded_empty(x, M, N2), _padded_empty(x, M, N1), _padded_empty(x, M, N1),
            _padded_empty(x, M, N2) if mul is not None else x.new_empty((0,), dtype=torch.float32))


mlp_gelu_fwd = _op("mlp_gelu_fwd", "(Tensor x, Tensor fc1_weight, Tensor? fc1_bias, Tensor fc2_weight, Tensor? fc2_bias, "
                   "Tensor? mul) -> (Tensor, Tensor, Tensor, Tensor)", _mlp_gelu_fwd_impl, _mlp_gelu_fwd_fake)


def _mlp_gelu_bwd_impl(dy, x, w1, w2, hidden, z, pre, mul, has_b1, has_b2, need_dx):
    """dW2/db2, then d(pre-activation) = (dY W2) * gelu'(z) with the GELU derivative fused into that GEMM's epilogue
    (NRM_EPI_DGELU), then dW1/db1 and dX -- no elementwise pass over the hidden activations in either direction.  With a
    multiplier (the gate), dY is first split by one fused kernel into d(fc2 output) = dY * mul and d(mul) = dY * fc2 output."""
    _require_gpu(dy, x, w1, w2)
    o1, o2 = w1, w2
    p1, p2 = w1, w2                                                     # the parameters the weight gradients are for
    x, w1, w2 = _rows(x), _f32c(w1), _f32c(w2)
    o1, o2 = (w1 if w1 is o1 else None), (w2 if w2 is o2 else None)
    N1, K1 = w1.shape
    N2, K2 = w2.shape
    dy = _rows(dy)
    dev = x.device
    M = x.shape[0]
    dmul = torch.empty((0,), dtype=torch.float32, device=dev)
    if mul is not None:
        m = _rows(mul)
        ld = _pad4(N2)
        if N2 % 4 == 0:
            dg = torch.empty(M, ld, dtype=torch.float32, device=dev)
            dmul_buf = torch.empty(M, ld, dtype=torch.float32, device=dev)
            native.call("nrm_mul_bwd", native.ptr(dy), dy.stride(0), native.ptr(pre), pre.stride(0), native.ptr(m), m.stride(0),
                        native.ptr(dg), native.ptr(dmul_buf), ld, M, N2, native.stream_ptr())
            dy, dmul = dg[:, :N2], dmul_buf[:, :N2]
        else:
            dy, dmul = _rows(dy * m), dy * pre
    dw2, db2 = _gemm_tn(dy, hidden, has_b2, target=p2)               # dW2[n,k] = sum_m dy[m,n] hidden[m,k]
    # d(pre-activation of fc1) = (dY W2) * gelu'(z): epilogue 2 reads z and writes the product
    dz, _ = _gemm_nt(dy, w2, 1, K2, K2, N2, None, EPI_DGELU, z=z, owner=o2)
    dw1, db1 = _gemm_tn(dz, x, has_b1, target=p1)
    e = lambda: torch.empty((0,), dtype=torch.float32, device=dev)      # noqa: E731
    dx = _gemm_nt(dz, w1, 1, K1, K1, N1, None, EPI_BIAS, owner=o1)[0] if need_dx else e()
    return dx, dw1, (db1 if has_b1 else e()), dw2, (db2 if has_b2 else e()), dmul


def _mlp_gelu_bwd_fake(dy, x, w1, w2, hidden, z, pre, mul, has_b1, has_b2, need_dx):
    M, (N1, K1), (N2, K2) = x.shape[0], w1.shape, w2.shape
    f = lambda *shape: x.new_empty(shape, dtype=torch.float32)          # noqa: E731
    return (_padded_empty(x, M, K1) if need_dx else f(0), f(N1, K1), f(N1) if has_b1 else f(0), f(N2, K2),
            f(N2) if has_b2 else f(0), (_padded_empty(x, M, N2) if N2 % 4 == 0 else f(M, N2)) if mul is not None else f(0))


mlp_gelu_bwd = _op("mlp_gelu_bwd", "(Tensor dy, Tensor x, Tensor fc1_weight, Tensor fc2_weight, Tensor hidden, Tensor z, "
                   "Tensor pre, Tensor? mul, bool has_b1, bool has_b2, bool need_dx) -> "
                   "(Tensor, Tensor, Tensor, Tensor, Tensor, Tensor)", _mlp_gelu_bwd_impl, _mlp_gelu_bwd_fake)


def _mlp_setup(ctx, inputs, output):
    x, w1, b1, w2, b2, mul = inputs
    y, hidden, z, pre = output
    ctx.set_materialize_grads(False)
    ctx.has_bias = (b1 is not None, b2 is not None)
    ctx.has_mul = mul is not None
    ctx.save_for_backward(x, w1, w2, hidden, z, pre, *((mul,) if mul is not None else ()))


def _mlp_backward(ctx, dy, _dh, _dz, _dpre):
    if dy is None:
        return None, None, None, None, None, None
    saved = ctx.saved_tensors
    x, w1, w2, hidden, z, pre = saved[:6]
    mul = saved[6] if ctx.has_mul else None
    need = ctx.needs_input_grad
    dx, dw1, db1, dw2, db2, dmul = mlp_gelu_bwd(dy, x, w1, w2, hidden, z, pre, mul, ctx.has_bias[0], ctx.has_bias[1], need[0])
    return ((dx if need[0] else None), dw1, (db1 if ctx.has_bias[0] else None), dw2, (db2 if ctx.has_bias[1] else None),
            (dmul if ctx.has_mul else None))


torch.library.register_autograd("nrm::mlp_gelu_fwd", _mlp_backward, setup_context=_mlp_setup, lib=_LIB)


def mlp_gelu(x, fc1_weight, fc1_bias, fc2_weight, fc2_bias, mul=None):
    """Linear -> exact GELU -> Linear on the last dimension of x (one fused autograd node); ``mul`` (same shape as the
    output) multiplies the result inside the second GEMM's epilogue."""
    lead = x.shape[:-1]
    if x.numel() == 0 and x.shape[-1] == fc1_weight.shape[1]:
        out = _degenerate((*lead, fc2_weight.shape[0]), x, fc1_weight, fc1_bias, fc2_weight, fc2_bias)
        return out if mul is None else out * mul
    _require_gpu(x, fc1_weight, fc2_weight)
    m2 = mul.reshape(-1, mul.shape[-1]) if mul is not None else None
    y = mlp_gelu_fwd(x.reshape(-1, x.shape[-1]), fc1_weight, fc1_bias, fc2_weight, fc2_bias, m2)[0]
    return y.reshape(*lead, fc2_weight.shape[0])


# ------------------------------------------------------------------------------------------------ BatchNorm1d
def _bn_stats_impl(x, running_mean, running_var, momentum, eps):
    """Train-mode statistics of the [R, N] rows: mean, rstd = 1/sqrt(biased var + eps) (two-pass variance), and the
    running-statistics update of nn.BatchNorm1d (unbiased variance, momentum) in place -- two column reductions and
    two finalisation launches, no ATen arithmetic."""
    _require_gpu(x, running_mean, running_var)
    x = _rows(x)
    R, N = x.shape
    ld = x.stride(0)
    st = native.stream_ptr()
    dev = x.device
    s = torch.zeros(2, N, dtype=torch.float32, device=dev)
    mean = torch.empty(N, dtype=torch.float32, device=dev)
    rstd = torch.empty(N, dtype=torch.float32, device=dev)
    native.call("nrm_colreduce", 0, native.ptr(x), None, None, None, native.ptr(s[0]), None, R, N, ld, st)
    native.call("nrm_bn_finalize", 0, native.ptr(s[0]), native.ptr(mean), native.ptr(running_mean), R, N, float(momentum), float(eps), st)
    native.call("nrm_colreduce", 1, native.ptr(x), None, native.ptr(mean), None, native.ptr(s[1]), None, R, N, ld, st)
    native.call("nrm_bn_finalize", 1, native.ptr(s[1]), native.ptr(rstd), native.ptr(running_var), R, N, float(momentum), float(eps), st)
    return mean, rstd


batch_norm_stats = _op("batch_norm_stats", "(Tensor x, Tensor(a!) running_mean, Tensor(b!) running_var, float momentum, float eps) -> "
                       "(Tensor, Tensor)", _bn_stats_impl,
                       lambda x, rm, rv, momentum, eps: (x.new_empty((x.shape[1],), dtype=torch.float32),
                                                         x.new_empty((x.shape[1],), dtype=torch.float32)))


def _bn_apply_impl(x, mean, rstd, weight, bias, batch_stats):
    _require_gpu(x, mean, rstd, weight, bias)
    x = _rows(x)
    R, N = x.shape
    ld = x.stride(0)
    w, b = _f32c(weight), _f32c(bias)
    y = torch.empty(R, ld, dtype=torch.float32, device=x.device)
    native.call("nrm_bn_apply", native.ptr(x), native.ptr(mean), native.ptr(rstd), native.ptr(w), native.ptr(b),
                native.ptr(y), R, N, ld, native.stream_ptr())
    return y[:, :N]


batch_norm_apply = _op("batch_norm_apply", "(Tensor x, Tensor mean, Tensor rstd, Tensor weight, Tensor bias, bool batch_stats) -> Tensor",
                       _bn_apply_impl, lambda x, mean, rstd, weight, bias, batch_stats: _padded_empty(x, x.shape[0], x.shape[1]))


def _bn_bwd_impl(dy, x, mean, rstd, weight, training, add=None):
    _require_gpu(dy, x)
    x = _rows(x)
    R, N = x.shape
    ld = x.stride(0)
    st = native.stream_ptr()
    dev = x.device
    if not (dy.dim() == 2 and dy.stride(1) == 1 and dy.stride(0) == ld and dy.dtype == torch.float32
            and dy.data_ptr() % 16 == 0):
        buf = torch.zeros(R, ld, dtype=torch.float32, device=dev)
        buf[:, :N] = dy
        dy = buf[:, :N]
    w = _f32c(weight)
    s0 = torch.zeros(N, dtype=torch.float32, device=dev)           # d beta, d gamma: two outputs, two allocations (outputs of
    s1 = torch.zeros(N, dtype=torch.float32, device=dev)           # an op must not share storage)
    native.call("nrm_colreduce", 2, native.ptr(x), native.ptr(dy), native.ptr(mean), native.ptr(rstd),
                native.ptr(s0), native.ptr(s1), R, N, ld, st)
    dx = torch.empty(R, ld, dtype=torch.float32, device=dev)
    if add is not None and not (add.dim() == 2 and add.stride(1) == 1 and add.stride(0) == ld and add.dtype == torch.float32
                                and add.data_ptr() % 16 == 0):
        buf = torch.zeros(R, ld, dtype=torch.float32, device=dev)
        buf[:, :N] = add
        add = buf[:, :N]
    native.call("nrm_bn_backward", native.ptr(x), native.ptr(dy), native.ptr(mean), native.ptr(rstd), native.ptr(w),
                native.ptr(s0), native.ptr(s1), native.ptr(add) if add is not None else None, native.ptr(dx), R, N, ld,
                1 if training else 0, st)
    return dx[:, :N], s1, s0


def _bn_bwd_fake(dy, x, mean, rstd, weight, training):
    R, N = x.shape
    return _padded_empty(x, R, N), x.new_empty((N,), dtype=torch.float32), x.new_empty((N,), dtype=torch.float32)


batch_norm_bwd = _op("batch_norm_bwd", "(Tensor dy, Tensor x, Tensor mean, Tensor rstd, Tensor weight, bool training) -> "
                     "(Tensor, Tensor, Tensor)", _bn_bwd_impl, _bn_bwd_fake)


def _bn_setup(ctx, inputs, output):
    x, mean, rstd, weight, bias, batch_stats = inputs
    ctx.batch_stats = batch_stats
    ctx.save_for_backward(x, mean, rstd, weight)


def _bn_backward(ctx, dy):
    # with batch statistics (training) mean and rstd are functions of x: their contribution is inside the dx formula of
    # nrm_bn_backward, so no gradient is sent to the mean / rstd inputs themselves
    x, mean, rstd, weight = ctx.saved_tensors
    dx, dgamma, dbeta = batch_norm_bwd(dy, x, mean, rstd, weight, ctx.batch_stats)
    return dx, None, None, dgamma, dbeta, None


torch.library.register_autograd("nrm::batch_norm_apply", _bn_backward, setup_context=_bn_setup, lib=_LIB)


def batch_norm(x, bn):
    """nn.BatchNorm1d(x) for a 2-D x through the HIP column kernels.  Non-default module configurations
    (no affine / no running stats / cumulative momentum) and widths that are not a multiple of 4 use the
    module itself."""
    if (x.dim() != 2 or x.shape[1] % 4 or not bn.affine or not bn.track_running_stats or bn.momentum is None):
        return bn(x)
    training = bn.training
    if x.shape[0] <= (1 if training else 0):
        return bn(x)          # 0/1 rows in training: nn.BatchNorm1d raises its ValueError; 0 rows in eval: empty result
    _require_gpu(x, bn.weight)
    if training:
        bn.num_batches_tracked.add_(1)
        mean, rstd = batch_norm_stats(x.detach(), bn.running_mean, bn.running_var, float(bn.momentum), float(bn.eps))
    else:
        mean, rstd = bn.running_mean.to(torch.float32), torch.rsqrt(bn.running_var.to(torch.float32) + bn.eps)
    return batch_norm_apply(x, mean, rstd, bn.weight, bn.bias, training)


# ------------------------------------------------------------------------------------------------ BN -> gate MLP -> product
def _gate_block_fwd_impl(x, mean, rstd, bn_w, bn_b, w1, b1, w2, b2, batch_stats):
    """gate(BatchNorm(x)) * x (reference models/user_model.py:32-33) as ONE autograd node: the rows x feed BatchNorm AND the
    product, so their two gradient contributions are joined inside the BatchNorm backward kernel instead of by a separate
    elementwise pass over [B*T, N]."""
    c = _bn_apply_impl(x, mean, rstd, bn_w, bn_b, batch_stats)
    y, hidden, z, pre = _mlp_gelu_fwd_impl(c, w1, b1, w2, b2, x)
    return y, c, hidden, z, pre


def _gate_block_fwd_fake(x, mean, rstd, bn_w, bn_b, w1, b1, w2, b2, batch_stats):
    M, N, N1 = x.shape[0], x.shape[1], w1.shape[0]
    return (_padded_empty(x, M, N), _padded_empty(x, M, N), _padded_empty(x, M, N1), _padded_empty(x, M, N1), _padded_empty(x, M, N))


gate_block_fwd = _op("gate_block_fwd", "(Tensor x, Tensor mean, Tensor rstd, Tensor bn_weight, Tensor bn_bias, Tensor fc1_weight, "
                     "Tensor? fc1_bias, Tensor fc2_weight, Tensor? fc2_bias, bool batch_stats) -> (Tensor, Tensor, Tensor, Tensor, Tensor)",
                     _gate_block_fwd_impl, _gate_block_fwd_fake)


def _gate_block_bwd_impl(dy, x, mean, rstd, bn_w, c, w1, w2, hidden, z, pre, has_b1, has_b2, batch_stats):
    dc, dw1, db1, dw2, db2, dmul = _mlp_gelu_bwd_impl(dy, c, w1, w2, hidden, z, pre, x, has_b1, has_b2, True)
    dx, dgamma, dbeta = _bn_bwd_impl(dc, x, mean, rstd, bn_w, batch_stats, add=dmul)
    return dx, dgamma, dbeta, dw1, db1, dw2, db2


def _gate_block_bwd_fake(dy, x, mean, rstd, bn_w, c, w1, w2, hidden, z, pre, has_b1, has_b2, batch_stats):
    M, N = x.shape
    (N1, K1), (N2, K2) = w1.shape, w2.shape
    f = lambda *shape: x.new_empty(shape, dtype=torch.float32)          # noqa: E731
    return _padded_empty(x, M, N), f(N), f(N), f(N1, K1), f(N1) if has_b1 else f(0), f(N2, K2), f(N2) if has_b2 else f(0)


gate_block_bwd = _op("gate_block_bwd", "(Tensor dy, Tensor x, Tensor mean, Tensor rstd, Tensor bn_weight, Tensor c, Tensor fc1_weight, "
                     "Tensor fc2_weight, Tensor hidden, Tensor z, Tensor pre, bool has_b1, bool has_b2, bool batch_stats) -> "
                     "(Tensor, Tensor, Tensor, Tensor, Tensor, Tensor, Tensor)", _gate_block_bwd_impl, _gate_block_bwd_fake)


def _gate_block_setup(ctx, inputs, output):
    x, mean, rstd, bn_w, bn_b, w1, b1, w2, b2, batch_stats = inputs
    y, c, hidden, z, pre = output
    ctx.set_materialize_grads(False)
    ctx.flags = (b1 is not None, b2 is not None, batch_stats)
    ctx.save_for_backward(x, mean, rstd, bn_w, c, w1, w2, hidden, z, pre)


def _gate_block_backward(ctx, dy, _dc, _dh, _dz, _dp):
    if dy is None:
        return (None,) * 10
    has_b1, has_b2, batch_stats = ctx.flags
    dx, dgamma, dbeta, dw1, db1, dw2, db2 = gate_block_bwd(dy, *ctx.saved_tensors, has_b1, has_b2, batch_stats)
    return dx, None, None, dgamma, dbeta, dw1, (db1 if has_b1 else None), dw2, (db2 if has_b2 else None), None


torch.library.register_autograd("nrm::gate_block_fwd", _gate_block_backward, setup_context=_gate_block_setup, lib=_LIB)


def gate_block(x, bn, fc1_weight, fc1_bias, fc2_weight, fc2_bias):
    """gate(bn(x)) * x for 2-D rows x (models/user_model.py:32-33; the gate is MLP with the default exact GELU).  Falls back to
    the separate ops for BatchNorm configurations / widths the column kernels do not take."""
    if (x.dim() != 2 or x.shape[1] % 4 or not bn.affine or not bn.track_running_stats or bn.momentum is None
            or x.shape[0] <= (1 if bn.training else 0) or fc2_weight.shape[0] != x.shape[1]):
        return mlp_gelu(batch_norm(x, bn), fc1_weight, fc1_bias, fc2_weight, fc2_bias, mul=x)
    _require_gpu(x, bn.weight, fc1_weight)
    if bn.training:
        bn.num_batches_tracked.add_(1)
        mean, rstd = batch_norm_stats(x.detach(), bn.running_mean, bn.running_var, float(bn.momentum), float(bn.eps))
    else:
        mean, rstd = bn.running_mean.to(torch.float32), torch.rsqrt(bn.running_var.to(torch.float32) + bn.eps)
    return gate_block_fwd(x, mean, rstd, bn.weight, bn.bias, fc1_weight, fc1_bias, fc2_weight, fc2_bias, bn.training)[0]


# ------------------------------------------------------------------------------------------------ concat
def _concat_cols_impl(parts):
    """cat(parts, dim=1) for 2-D fp32 row blocks in one launch (models/user_model.py:31, user_invariant_interest_model.py:81,88)."""
    import ctypes
    _require_gpu(*parts)
    parts = [p if (p.dtype == torch.float32 and p.dim() == 2 and p.stride(1) == 1) else _f32c(p) for p in parts]
    R = parts[0].shape[0]
    total = sum(p.shape[1] for p in parts)
    if len(parts) > 8 or any(p.shape[0] != R for p in parts):
        return torch.cat(parts, dim=1)
    out = torch.empty(R, _pad4(total), dtype=torch.float32, device=parts[0].device)
    k = len(parts)
    srcs = (ctypes.c_void_p * k)(*[p.data_ptr() for p in parts])
    lds = (ctypes.c_long * k)(*[p.stride(0) if p.shape[0] > 1 else p.shape[1] for p in parts])
    widths = (ctypes.c_int * k)(*[p.shape[1] for p in parts])
    native.call("nrm_concat_cols", srcs, lds, widths, k, native.ptr(out), out.stride(0), R, native.stream_ptr())
    return out[:, :total]


concat_cols_op = _op("concat_cols", "(Tensor[] parts) -> Tensor", _concat_cols_impl,
                     lambda parts: _padded_empty(parts[0], parts[0].shape[0], sum(p.shape[1] for p in parts)))


def _concat_setup(ctx, inputs, output):
    ctx.widths = [p.shape[1] for p in inputs[0]]


def _concat_backward(ctx, g):
    out, c = [], 0
    for w in ctx.widths:
        out.append(g[:, c:c + w])
        c += w
    return (out,)


torch.library.register_autograd("nrm::concat_cols", _concat_backward, setup_context=_concat_setup, lib=_LIB)


def concat_last(parts):
    """torch.cat(parts, dim=-1) for tensors that agree in their leading dimensions, as one HIP launch."""
    lead = parts[0].shape[:-1]
    rows = 1
    for d in lead:
        rows *= d
    if rows == 0 or len(parts) > 8:
        return torch.cat(parts, dim=-1)
    _require_gpu(*parts)
    y = concat_cols_op([p.reshape(rows, p.shape[-1]) for p in parts])
    return y.reshape(*lead, y.shape[-1]) if y.is_contiguous() else y.unflatten(0, tuple(lead))


# ------------------------------------------------------------------------------------------------ pool
def _pool_fwd_impl(s, h):
    """pooled[b,t,:] = sum_h s[b,t,h] * h[b,h,:]  (un-normalised, unmasked: reference
    models/user_invariant_interest_model.py:86-87)."""
    _require_gpu(s, h)
    s, h = _f32c(s), _f32c(h)
    B, T, H = s.shape
    D = h.shape[2]
    out = torch.empty(B, T, D, dtype=torch.float32, device=h.device)
    native.call("nrm_pool_bmm", native.ptr(s), T * H, H, 1, native.ptr(h), D, native.ptr(out), B, T, H, D, 0,
                native.stream_ptr())
    return out


weighted_pool_fwd = _op("weighted_pool_fwd", "(Tensor scores, Tensor history) -> Tensor", _pool_fwd_impl,
                        lambda s, h: s.new_empty((s.shape[0], s.shape[1], h.shape[2]), dtype=torch.float32))


def _pooled_grad_rows(g, B, T, D):
    """The pooled gradient [B,T,D] as (tensor, row stride): a column block of a wider row-major matrix (the head gradient's
    slice, as concat's backward hands it over) is read in place; anything else is made contiguous."""
    g = g if g.dtype == torch.float32 else g.to(torch.float32)
    ld = g.stride(1) if T > 1 else (g.stride(0) if B > 1 else D)
    if (g.dim() == 3 and g.stride(2) == 1 and ld >= D and ld % 4 == 0 and (B == 1 or g.stride(0) == T * ld)
            and (T == 1 or g.stride(1) == ld) and g.data_ptr() % 16 == 0 and T * ld * 4 < (1 << 31)):
        return g, ld
    return g.contiguous(), D


def _pool_bwd_impl(g, s, h):
    _require_gpu(g, s, h)
    s, h = _f32c(s), _f32c(h)
    B, T, H = s.shape
    D = h.shape[2]
    g, ldg = _pooled_grad_rows(g, B, T, D)
    st = native.stream_ptr()
    ds = torch.empty(B, T, H, dtype=torch.float32, device=h.device)
    native.call("nrm_pool_rowdot", native.ptr(g), ldg, native.ptr(h), native.ptr(ds), B, T, H, D, None, 0, st)
    dh = torch.empty(B, H, D, dtype=torch.float32, device=h.device)
    native.call("nrm_pool_bmm", native.ptr(s), T * H, 1, H, native.ptr(g), ldg, native.ptr(dh), B, H, T, D, 0, st)
    return ds, dh


weighted_pool_bwd = _op("weighted_pool_bwd", "(Tensor g, Tensor scores, Tensor history) -> (Tensor, Tensor)", _pool_bwd_impl,
                        lambda g, s, h: (s.new_empty(tuple(s.shape), dtype=torch.float32), h.new_empty(tuple(h.shape), dtype=torch.float32)))


def _pool_setup(ctx, inputs, output):
    ctx.save_for_backward(*inputs)


def _pool_backward(ctx, g):
    s, h = ctx.saved_tensors
    return weighted_pool_bwd(g, s, h)


torch.library.register_autograd("nrm::weighted_pool_fwd", _pool_backward, setup_context=_pool_setup, lib=_LIB)


def weighted_pool(scores, history):
    D = history.shape[-1]
    if scores.numel() == 0:                                  # empty batch / no candidates / empty history: sum of nothing
        return _degenerate((scores.shape[0], scores.shape[1], D), scores, history)
    _require_gpu(scores, history)
    if D % 4 == 0:
        return weighted_pool_fwd(scores, history)
    return weighted_pool_fwd(scores, torch.nn.functional.pad(history.to(torch.float32), (0, _pad4(D) - D)))[..., :D]


# ------------------------------------------------------------------------------------------------ attention + pool, one node
def _attend_pool_fwd_impl(t, h, w1, b1, w2, b2, save_z, mma):
    """pooled[b,t,:] = sum_h s[b,t,h] h[b,h,:] with s = the pointwise attention scores: the two calls of
    user_invariant_interest_model.py:83-87 as ONE autograd node, so that its backward can chain the pool's and the
    attention's kernels (see _attend_pool_bwd_impl)."""
    s, z = _pwattn_fwd_impl(t, h, w1, b1, w2, b2, save_z, mma)
    return _pool_fwd_impl(s, h), s, z


def _attend_pool_fwd_fake(t, h, w1, b1, w2, b2, save_z, mma):
    s, z = _pwattn_fwd_fake(t, h, w1, b1, w2, b2, save_z, mma)
    return t.new_empty(tuple(t.shape), dtype=torch.float32), s, z


attend_pool_fwd = _op("attend_pool_fwd", "(Tensor t, Tensor h, Tensor fc1_weight, Tensor fc1_bias, Tensor fc2_weight, Tensor fc2_bias, "
                      "bool save_z, int mma) -> (Tensor, Tensor, Tensor)", _attend_pool_fwd_impl, _attend_pool_fwd_fake)


def _attend_pool_bwd_impl(g, t, h, w1, w2, s, z, mma, need_dt, need_dh):
    """Backward of pool + attention from the pooled gradient g [B,T,D] (read in place when it is a column block of the head
    gradient).  Compared with the two separate nodes: the pool's rowdot launch also clears the attention's dw2 | db2
    accumulators, and the pool's history gradient is ADDED onto the attention's by the last launch -- no fill, no [B,T,D]
    copy of g, no [B,H,D] add by autograd."""
    _require_gpu(g, t, h, w1, w2, s, z)
    B, T, D = t.shape
    H = h.shape[1]
    s, h = _f32c(s), _f32c(h)
    g, ldg = _pooled_grad_rows(g, B, T, D)
    st = native.stream_ptr()
    dev = h.device
    acc = torch.empty(D + 4, dtype=torch.float32, device=dev)
    ds = torch.empty(B, T, H, dtype=torch.float32, device=dev)
    native.call("nrm_pool_rowdot", native.ptr(g), ldg, native.ptr(h), native.ptr(ds), B, T, H, D, native.ptr(acc), D + 4, st)
    dt, dh, dw1, db1, dw2b2 = _attn_bwd_core(ds, t, h, w1, w2, z, mma, need_dt, need_dh, acc=acc)
    if need_dh:
        native.call("nrm_pool_bmm", native.ptr(s), T * H, 1, H, native.ptr(g), ldg, native.ptr(dh), B, H, T, D, 1, st)
    return dt, dh, dw1, db1, dw2b2


attend_pool_bwd = _op("attend_pool_bwd", "(Tensor g, Tensor t, Tensor h, Tensor fc1_weight, Tensor fc2_weight, Tensor s, Tensor(a!) z, "
                      "int mma, bool need_dt, bool need_dh) -> (Tensor, Tensor, Tensor, Tensor, Tensor)", _attend_pool_bwd_impl,
                      lambda g, t, h, w1, w2, s, z, mma, need_dt, need_dh: _pwattn_bwd_fake(s, t, h, w1, w2, z, mma, need_dt, need_dh))


def _attend_pool_setup(ctx, inputs, output):
    t, h, w1, b1, w2, b2, save_z, mma = inputs
    pooled, s, z = output
    ctx.set_materialize_grads(False)
    ctx.save_z = save_z
    ctx.mma = mma
    ctx.consumed = False
    ctx.w2_shape, ctx.b2_shape = tuple(w2.shape), tuple(b2.shape)
    ctx.mark_non_differentiable(s, z)
    if save_z:
        ctx.save_for_backward(t, h, w1, w2, s, z, b1)


def _attend_pool_backward(ctx, g, _ds, _dz):
    if g is None:
        return None, None, None, None, None, None, None, None
    if not ctx.save_z:
        raise RuntimeError("pointwise attention: the forward ran with save_z=False (no-grad / inference call); there "
                           "is nothing to differentiate through")
    t, h, w1, w2, s, z, b1 = ctx.saved_tensors
    need = ctx.needs_input_grad
    dt, dh, dw1, db1, dw2b2 = attend_pool_bwd(g, t, h, w1, w2, s, _consume_z(ctx, z, t, h, w1, b1, w2), ctx.mma, bool(need[0]), bool(need[1]))
    D = t.shape[2]
    return ((dt if need[0] else None), (dh if need[1] else None), dw1, db1, dw2b2[:D].reshape(ctx.w2_shape),
            dw2b2[D:D + 1].reshape(ctx.b2_shape), None, None)


torch.library.register_autograd("nrm::attend_pool_fwd", _attend_pool_backward, setup_context=_attend_pool_setup, lib=_LIB)


def attend_and_pool(target, history, fc1_weight, fc1_bias, fc2_weight, fc2_bias, mma=None):
    """[B,T,D] x [B,H,D] -> pooled [B,T,D] = weighted_pool(pointwise_attention_scores(...), history) as one autograd node."""
    D = target.shape[-1]
    if D % 4 or target.dim() != 3 or target.shape[0] * target.shape[1] * history.shape[1] == 0:
        return weighted_pool(pointwise_attention_scores(target, history, fc1_weight, fc1_bias, fc2_weight, fc2_bias, mma=mma), history)
    args = (target, history, fc1_weight, fc1_bias, fc2_weight, fc2_bias)
    _require_gpu(*args)
    save_z = torch.is_grad_enabled() and any(a.requires_grad for a in args)
    return attend_pool_fwd(*args, save_z, resolve_mma(mma))[0]


# ------------------------------------------------------------------------------------------------ loss
_index_error_flag = {}


# z elements (B*T*H*D) of one attention up to which UserInvariantInterestModel runs its two attentions on two streams.  Rounds
# 1-2 kept C3 (614 M) and C5 on one stream because bench.py timed its kernels inside the timed region; it now takes kernel
# durations from separate one-stream steps, and with the text+image branch's backward reduced to dz + the dW_p pass (DESIGN.md
# section 4d) the HBM-bound dz pass of one branch hides under the MFMA-bound contraction of the other: C3 32.2 -> 31.95 ms, C5
# 119.0 -> 117.3 ms.  So: every size.  (NRM_BRANCH_STREAMS=0 or module.two_streams = False for one stream.)
BRANCH_STREAMS_MAX_ELEMS = 1 << 62
_branch_streams = {}


def branch_stream(like):
    """The side stream of ``like``'s device (one per device and process, created on first use)."""
    key = like.device.index if like.device.index is not None else torch.cuda.current_device()
    st = _branch_streams.get(key)
    if st is None:
        st = _branch_streams[key] = torch.cuda.Stream(device=like.device)
    return st


def index_error_flag(device):
    """int32 flag set to 1 by the front-end / loss kernels when a packed row holds an out-of-range table index or a user id
    outside delta (the reference raises IndexError there; the kernels clamp, flag and go on).  It lives in PINNED HOST memory
    that the device writes directly (zero-copy; written only in the error case), so the host can look at it at any time without
    a copy or a synchronisation: ``trainer.IndexErrorWatch`` does at the start of every step, ``check_index_errors``
    synchronises first and is therefore exact.  One flag per device."""
    device = torch.device(device)
    if device.index is None:
        device = torch.device(device.type, torch.cuda.current_device())
    key = str(device)
    if key not in _index_error_flag:
        _index_error_flag[key] = torch.zeros(1, dtype=torch.int32).pin_memory()
    return _index_error_flag[key]


def check_index_errors(device="cuda"):
    """Wait for the device, then turn a raised flag into the reference's IndexError (and clear it)."""
    device = torch.device(device)
    flag = index_error_flag(device)
    torch.cuda.synchronize(device if device.index is not None else None)
    if int(flag[0]):
        flag.zero_()
        raise IndexError("index out of range (a category / type / time table index of a packed feature row, or a user id "
                         "outside delta)")


def _loss_impl(out, delta, label, user_id, alpha):
    """The two-term BCE-on-softmax loss of reference models/user_model.py:37-43, value and gradients in one
    kernel (one wave per impression).  Returns (loss, dout, ddelta); dout [B,T] = dL/dout is column 0 of a zero-padded
    [B*T, 4] matrix -- the layout the GEMM that consumes it streams, so nothing re-pads it."""
    _require_gpu(out, delta, label, user_id)
    B, T = out.shape
    d = _f32c(delta)
    # the logits arrive as column 0 of the padded [B*T, 4] output of the last GEMM (strides (4T, 4)): read in place
    so = out.stride(1) if T > 1 else (out.stride(0) if B > 1 else 1)
    if not (out.dtype == torch.float32 and so >= 1 and (T == 1 or out.stride(1) == so) and (B == 1 or out.stride(0) == T * so)):
        out, so = _f32c(out), 1
    y = label if (label.dtype in (torch.float32, torch.float64) and label.is_contiguous()) else _f32c(label)
    uid = user_id.to(torch.int64).contiguous()
    loss = torch.zeros((), dtype=torch.float32, device=out.device)
    ddelta = torch.zeros_like(d)
    dbuf = torch.empty(B * T, 4, dtype=torch.float32, device=out.device)
    native.call("nrm_loss_fwd_bwd", native.ptr(out), so, native.ptr(y), 1 if y.dtype == torch.float64 else 0, native.ptr(uid),
                native.ptr(d), d.numel(), float(alpha), B, T, native.ptr(loss), native.ptr(dbuf), 4, native.ptr(ddelta),
                native.ptr(index_error_flag(out.device)), native.stream_ptr())
    return loss, dbuf[:, 0].view(B, T), ddelta


softmax_bce_loss_op = _op("softmax_bce_loss", "(Tensor out, Tensor delta, Tensor label, Tensor user_id, float alpha) -> "
                          "(Tensor, Tensor, Tensor)", _loss_impl,
                          lambda out, delta, label, user_id, alpha: (out.new_empty((), dtype=torch.float32),
                                                                     out.new_empty((out.shape[0] * out.shape[1], 4), dtype=torch.float32)[:, 0]
                                                                     .view(out.shape[0], out.shape[1]),
                                                                     delta.new_empty(tuple(delta.shape), dtype=torch.float32)))


def _loss_setup(ctx, inputs, output):
    ctx.set_materialize_grads(False)
    ctx.save_for_backward(output[1], output[2])


# The gradient train_step seeds the backward with: a cached scalar 1.  The loss backward recognises it (by address) and hands
# its saved dL/dout and dL/ddelta on as they are -- no ones_like fill by autograd, no two multiplies by one.
_unit_grads = {}


def unit_grad(like):
    """The cached scalar 1 of ``like``'s device -- or None while a stream capture is under way and none exists yet (a tensor
    created inside a capture gets its value only when the graph is replayed; backward(None) then lets autograd seed itself)."""
    key = like.device.index if like.device.index is not None else torch.cuda.current_device()
    u = _unit_grads.get(key)
    if u is None:
        if torch.cuda.is_current_stream_capturing():
            return None
        u = _unit_grads[key] = torch.ones((), dtype=torch.float32, device=like.device)
    return u


def _loss_backward(ctx, gl, _a, _b):
    if gl is None:
        return None, None, None, None, None
    dout, ddelta = ctx.saved_tensors
    u = _unit_grads.get(dout.device.index)
    try:
        unit = u is not None and gl.dim() == 0 and gl.data_ptr() == u.data_ptr()
    except RuntimeError:                                  # fake / functional tensors (tracing) have no address: the general formula
        unit = False
    if unit:
        # dout: the saved buffer itself (a strided view: AccumulateGrad never adopts it).  ddelta: a COPY -- AccumulateGrad adopts a
        # contiguous gradient as delta.grad and a later in-place accumulation would rewrite what a second backward through a
        # retained graph reads (user_num + 1 floats: one tiny launch)
        return dout, ddelta.clone(), None, None, None
    return dout * gl, ddelta * gl, None, None, None


torch.library.register_autograd("nrm::softmax_bce_loss", _loss_backward, setup_context=_loss_setup, lib=_LIB)


def softmax_bce_loss(out, delta, label, user_id, alpha):
    if out.numel() == 0:                                     # nn.BCELoss: mean over no elements
        return _degenerate((), out, delta) + float("nan")
    _require_gpu(out, delta, label, user_id)
    return softmax_bce_loss_op(out, delta, label, user_id, float(alpha))[0]


# ------------------------------------------------------------------------------------------------ embedding front end
def _frontend_dims(cat_tab, sen_w, type_tab, year_tab, month_tab, day_tab, hour_tab):
    return (cat_tab.shape[0], cat_tab.shape[1], sen_w.shape[0], type_tab.shape[0], type_tab.shape[1], year_tab.shape[0],
            month_tab.shape[0], day_tab.shape[0], hour_tab.shape[0], year_tab.shape[1])


def _frontend_fwd_impl(x, behaviour, n_sub, P, cat_tab, sen_w, sen_b, type_tab, year_tab, month_tab, day_tab, hour_tab):
    """Packed rows [R, cols] (fp32 or fp64) -> (label rows [R, e0+e1+e2+e3(+2)], text/image rows [R, P] fp32)."""
    _require_gpu(x, cat_tab)
    if x.dtype not in (torch.float32, torch.float64):
        x = x.to(torch.float32)
    x = x.contiguous()
    R, xcols = x.shape
    tabs = [_f32c(t) for t in (cat_tab, sen_w, sen_b, type_tab, year_tab, month_tab, day_tab, hour_tab)]
    cat_tab, sen_w, sen_b, type_tab, year_tab, month_tab, day_tab, hour_tab = tabs
    dims = _frontend_dims(cat_tab, sen_w, type_tab, year_tab, month_tab, day_tab, hour_tab)
    n_cat, e0, e1, n_type, e2, n_year, n_month, n_day, n_hour, e3 = dims
    width = e0 + e1 + e2 + e3 + (2 if behaviour else 0)
    ldlab, ldti = _pad4(width), _pad4(P)
    lab = torch.empty(R, ldlab, dtype=torch.float32, device=x.device)
    ti = torch.empty(R, ldti, dtype=torch.float32, device=x.device)
    native.call("nrm_frontend_fwd", native.ptr(x), 1 if x.dtype == torch.float64 else 0, R, xcols, P, n_sub,
                1 if behaviour else 0, native.ptr(cat_tab), n_cat, e0, native.ptr(sen_w), native.ptr(sen_b), e1,
                native.ptr(type_tab), n_type, e2, native.ptr(year_tab), native.ptr(month_tab), native.ptr(day_tab),
                native.ptr(hour_tab), n_year, n_month, n_day, n_hour, e3,
                native.ptr(lab), ldlab, native.ptr(ti), ldti, native.ptr(index_error_flag(x.device)),
                native.stream_ptr())
    return lab[:, :width], ti[:, :P]


def _frontend_fwd_fake(x, behaviour, n_sub, P, cat_tab, sen_w, sen_b, type_tab, year_tab, month_tab, day_tab, hour_tab):
    width = cat_tab.shape[1] + sen_w.shape[0] + type_tab.shape[1] + year_tab.shape[1] + (2 if behaviour else 0)
    return _padded_empty(x, x.shape[0], width), _padded_empty(x, x.shape[0], P)


_FRONT_TABLES = "Tensor cat_tab, Tensor sen_w, Tensor sen_b, Tensor type_tab, Tensor year_tab, Tensor month_tab, Tensor day_tab, Tensor hour_tab"
frontend_fwd = _op("frontend_fwd", f"(Tensor x, bool behaviour, int n_sub, int P, {_FRONT_TABLES}) -> (Tensor, Tensor)",
                   _frontend_fwd_impl, _frontend_fwd_fake)


def _frontend_bwd_impl(dlab, x, behaviour, n_sub, P, cat_tab, sen_w, sen_b, type_tab, year_tab, month_tab, day_tab, hour_tab):
    tabs = (cat_tab, sen_w, sen_b, type_tab, year_tab, month_tab, day_tab, hour_tab)
    return _frontend_bwd_sets([(dlab, x, behaviour)], n_sub, P, tabs)


# category-table gradient by counting sort + gather-sum (frontend.hip: cat_grad_launch) instead of one float atomic per table
# reference and column -- from FE_SORT_MIN_ATOMICS atomics on: the sort's five small launches cost more than they save below
# (measured per step: C3, 98 M atomics: 32.4 -> 31.9 ms; C2, 24 M: 3.36 -> 3.39; reference default, 11 M: 1.51 -> 1.65).
# NRM_FE_SORT=0 / 1 forces the scatter / the sort.
FE_SORT_MIN_ATOMICS = 50_000_000


def _frontend_bwd_sets(sets, n_sub, P, tabs):
    """Backward of the front end for one or two row sets [(dlab, x, behaviour)] into ONE zeroed arena (returned)."""
    cat_tab, sen_w, sen_b, type_tab, year_tab, month_tab, day_tab, hour_tab = tabs
    dev = sets[0][1].device
    sen_w, sen_b = _f32c(sen_w), _f32c(sen_b)
    dims = _frontend_dims(cat_tab, sen_w, type_tab, year_tab, month_tab, day_tab, hour_tab)
    n_cat, e0, e1, n_type, e2, n_year, n_month, n_day, n_hour, e3 = dims
    grads, arena = _frontend_grad_arena(tabs, dev)
    d_cat, d_sw, d_sb, d_type, d_year, d_month, d_day, d_hour = grads
    prepared = []
    for dlab, x, behaviour in sets:
        _require_gpu(dlab, x)
        if x.dtype not in (torch.float32, torch.float64):
            x = x.to(torch.float32)
        prepared.append((_rows(dlab), x.contiguous(), behaviour))
    refs = sum(x.shape[0] for _, x, _ in prepared) * (n_sub + 1)
    forced = _os.environ.get("NRM_FE_SORT")
    sort = (forced != "0" and (forced == "1" or refs * e0 >= FE_SORT_MIN_ATOMICS) and 1 <= n_sub <= 16 and e0 <= 512
            and len({x.dtype for _, x, _ in prepared}) == 1 and refs < (1 << 31))
    st = native.stream_ptr()
    for dlab, x, behaviour in prepared:
        native.call("nrm_frontend_bwd", native.ptr(x), 1 if x.dtype == torch.float64 else 0, x.shape[0], x.shape[1], P,
                    n_sub, 1 if behaviour else 0, native.ptr(dlab), dlab.stride(0), native.ptr(sen_w), native.ptr(sen_b),
                    n_cat, e0, e1, n_type, e2, n_year, n_month, n_day, n_hour, e3,
                    None if sort else native.ptr(d_cat), native.ptr(d_sw), native.ptr(d_sb), native.ptr(d_type),
                    native.ptr(d_year), native.ptr(d_month), native.ptr(d_day), native.ptr(d_hour), st)
    if sort:
        (dl0, x0, _), (dl1, x1, _) = prepared[0], (prepared[1] if len(prepared) > 1 else (None, None, None))
        rows = x0.shape[0] + (x1.shape[0] if x1 is not None else 0)
        ws = torch.empty(native.load().nrm_frontend_cat_ws_ints(n_cat, rows, n_sub), dtype=torch.int32, device=dev)
        native.call("nrm_frontend_cat_grad", native.ptr(x0), x0.shape[0], x0.shape[1], native.ptr(dl0), dl0.stride(0),
                    native.ptr(x1) if x1 is not None else None, x1.shape[0] if x1 is not None else 0,
                    x1.shape[1] if x1 is not None else 0, native.ptr(dl1) if x1 is not None else None,
                    dl1.stride(0) if x1 is not None else 0, 1 if x0.dtype == torch.float64 else 0, P, n_sub, n_cat, e0,
                    native.ptr(d_cat), native.ptr(ws), st)
    return arena


def _frontend_arena_floats(tabs):
    return sum(_pad4(int(t.numel())) for t in tabs)


def _frontend_bwd_fake(dlab, x, behaviour, n_sub, P, cat_tab, sen_w, sen_b, type_tab, year_tab, month_tab, day_tab, hour_tab):
    tabs = (cat_tab, sen_w, sen_b, type_tab, year_tab, month_tab, day_tab, hour_tab)
    return cat_tab.new_empty((_frontend_arena_floats(tabs),), dtype=torch.float32)


# returns ONE buffer holding the eight table gradients back to back (each padded to 4 floats): _frontend_grad_arena(tabs, dev, arena)
# gives the eight views (an op's outputs must not share storage, so the views are taken outside the op)
frontend_bwd = _op("frontend_bwd", f"(Tensor dlab, Tensor x, bool behaviour, int n_sub, int P, {_FRONT_TABLES}) -> Tensor",
                   _frontend_bwd_impl, _frontend_bwd_fake)


def _frontend_setup(ctx, inputs, output):
    ctx.set_materialize_grads(False)
    ctx.args = inputs[1:4]
    ctx.save_for_backward(inputs[0], *inputs[4:])


def _frontend_backward(ctx, dlab, _dti):
    if dlab is None:
        return (None,) * 12
    x, *tabs = ctx.saved_tensors
    arena = frontend_bwd(dlab, x, *ctx.args, *tabs)
    return (None, None, None, None) + _frontend_grad_arena(tabs, arena.device, arena)[0]


torch.library.register_autograd("nrm::frontend_fwd", _frontend_backward, setup_context=_frontend_setup, lib=_LIB)


def frontend(x, behaviour, n_sub, P, cat_tab, sen_w, sen_b, type_tab, year_tab, month_tab, day_tab, hour_tab):
    """x [B, N, cols] -> (label rows [B, N, width], text/image rows [B, N, P])."""
    B, N = x.shape[0], x.shape[1]
    tabs = (cat_tab, sen_w, sen_b, type_tab, year_tab, month_tab, day_tab, hour_tab)
    if B * N == 0:
        width = cat_tab.shape[1] + sen_w.shape[0] + type_tab.shape[1] + year_tab.shape[1] + (2 if behaviour else 0)
        return _degenerate((B, N, width), x, *tabs), _degenerate((B, N, int(P)), x)
    _require_gpu(x, cat_tab)
    lab, ti = frontend_fwd(x.reshape(B * N, x.shape[2]), bool(behaviour), int(n_sub), int(P), *tabs)
    ti = ti.detach()                                            # a copy of input columns: not differentiable
    return lab.reshape(B, N, -1) if lab.is_contiguous() else lab.unflatten(0, (B, N)), ti.unflatten(0, (B, N))


# history + candidate rows through ONE autograd node: the two backward launches accumulate (float atomics) into one zeroed
# arena, instead of two arenas, two fills and eight adds of autograd
def _frontend_pair_fwd_impl(xh, xt, n_sub, P, cat_tab, sen_w, sen_b, type_tab, year_tab, month_tab, day_tab, hour_tab):
    tabs = (cat_tab, sen_w, sen_b, type_tab, year_tab, month_tab, day_tab, hour_tab)
    lab_h, ti_h = _frontend_fwd_impl(xh, True, n_sub, P, *tabs)
    lab_t, ti_t = _frontend_fwd_impl(xt, False, n_sub, P, *tabs)
    return lab_h, ti_h, lab_t, ti_t


def _frontend_pair_fwd_fake(xh, xt, n_sub, P, *tabs):
    return _frontend_fwd_fake(xh, True, n_sub, P, *tabs) + _frontend_fwd_fake(xt, False, n_sub, P, *tabs)


frontend_pair_fwd = _op("frontend_pair_fwd", f"(Tensor x_history, Tensor x_target, int n_sub, int P, {_FRONT_TABLES}) -> "
                        "(Tensor, Tensor, Tensor, Tensor)", _frontend_pair_fwd_impl, _frontend_pair_fwd_fake)


def _frontend_pair_bwd_impl(dlab_h, dlab_t, xh, xt, n_sub, P, cat_tab, sen_w, sen_b, type_tab, year_tab, month_tab, day_tab, hour_tab):
    tabs = (cat_tab, sen_w, sen_b, type_tab, year_tab, month_tab, day_tab, hour_tab)
    sets = [(dlab, x, beh) for dlab, x, beh in ((dlab_h, xh, True), (dlab_t, xt, False)) if dlab is not None]
    if not sets:
        return _frontend_grad_arena(tabs, xh.device)[1]
    return _frontend_bwd_sets(sets, n_sub, P, tabs)


def _frontend_grad_arena(tabs, dev, arena=None):
    """The eight gradient tables carved out of ONE zero-initialised buffer (one fill launch instead of eight)."""
    cat_tab, sen_w, sen_b, type_tab, year_tab, month_tab, day_tab, hour_tab = tabs
    shapes = [tuple(cat_tab.shape), tuple(sen_w.shape), tuple(sen_b.shape), tuple(type_tab.shape), tuple(year_tab.shape),
              tuple(month_tab.shape), tuple(day_tab.shape), tuple(hour_tab.shape)]
    sizes = [_pad4(int(torch.Size(sh).numel())) for sh in shapes]
    if arena is None:
        arena = torch.zeros(sum(sizes), dtype=torch.float32, device=dev)
    grads, o = [], 0
    for sh, n in zip(shapes, sizes):
        grads.append(arena[o:o + int(torch.Size(sh).numel())].view(sh))
        o += n
    return tuple(grads), arena


frontend_pair_bwd = _op("frontend_pair_bwd", f"(Tensor? dlab_history, Tensor? dlab_target, Tensor x_history, Tensor x_target, int n_sub, int P, "
                        f"{_FRONT_TABLES}) -> Tensor", _frontend_pair_bwd_impl,
                        lambda dh, dt, xh, xt, n_sub, P, *tabs: tabs[0].new_empty((_frontend_arena_floats(tabs),), dtype=torch.float32))


def _frontend_pair_setup(ctx, inputs, output):
    ctx.set_materialize_grads(False)
    ctx.args = inputs[2:4]
    ctx.save_for_backward(inputs[0], inputs[1], *inputs[4:])


def _frontend_pair_backward(ctx, dlab_h, _dti_h, dlab_t, _dti_t):
    if dlab_h is None and dlab_t is None:
        return (None,) * 12
    xh, xt, *tabs = ctx.saved_tensors
    arena = frontend_pair_bwd(dlab_h, dlab_t, xh, xt, *ctx.args, *tabs)
    return (None, None, None, None) + _frontend_grad_arena(tabs, arena.device, arena)[0]


torch.library.register_autograd("nrm::frontend_pair_fwd", _frontend_pair_backward, setup_context=_frontend_pair_setup, lib=_LIB)


def frontend_pair(x_history, x_target, n_sub, P, cat_tab, sen_w, sen_b, type_tab, year_tab, month_tab, day_tab, hour_tab):
    """(x_history [B,H,cols], x_target [B,T,cols]) -> (label rows of the history [B,H,width+2], its text/image rows [B,H,P], label rows
    of the candidates [B,T,width], their text/image rows [B,T,P]): frontend(x_history, True) and frontend(x_target, False) as one node."""
    tabs = (cat_tab, sen_w, sen_b, type_tab, year_tab, month_tab, day_tab, hour_tab)
    B, H, T = x_history.shape[0], x_history.shape[1], x_target.shape[1]
    if B * H == 0 or B * T == 0 or x_target.shape[0] != B:
        return frontend(x_history, True, n_sub, P, *tabs) + frontend(x_target, False, n_sub, P, *tabs)
    _require_gpu(x_history, x_target, cat_tab)
    lab_h, ti_h, lab_t, ti_t = frontend_pair_fwd(x_history.reshape(B * H, x_history.shape[2]), x_target.reshape(B * T, x_target.shape[2]),
                                                 int(n_sub), int(P), *tabs)
    shape = lambda a, n: a.reshape(B, n, -1) if a.is_contiguous() else a.unflatten(0, (B, n))      # noqa: E731
    return shape(lab_h, H), ti_h.detach().unflatten(0, (B, H)), shape(lab_t, T), ti_t.detach().unflatten(0, (B, T))


# ------------------------------------------------------------------------------------------------ evaluation / optimizer
def _row_auc_impl(score, label, length):
    """Per-impression ROC-AUC and top-1 hit (reference train.py:77-80, verify.py:25-36)."""
    _require_gpu(score, label, length)
    s, y = _f32c(score), _f32c(label)
    B, T = s.shape
    auc = torch.empty(B, dtype=torch.float32, device=s.device)
    top1 = torch.empty(B, dtype=torch.int32, device=s.device)
    ln = length.to(torch.int32).contiguous() if length is not None else None
    native.call("nrm_row_auc", native.ptr(s), native.ptr(y), native.ptr(ln) if ln is not None else None, B, T,
                native.ptr(auc), native.ptr(top1), native.stream_ptr())
    return auc, top1


row_auc = _op("row_auc", "(Tensor score, Tensor label, Tensor? length) -> (Tensor, Tensor)", _row_auc_impl,
              lambda score, label, length: (score.new_empty((score.shape[0],), dtype=torch.float32),
                                            score.new_empty((score.shape[0],), dtype=torch.int32)))


def _adam_step_impl(param, grad, exp_avg, exp_avg_sq, state, lr, beta1, beta2, eps, weight_decay, zero_grad):
    """train.py:48,73-75 over one flat buffer: Adam(lr, weight_decay) + optional zero_grad in one launch; the step
    counter lives in ``state`` on the device (hipGraph-capturable)."""
    _require_gpu(param, grad, exp_avg, exp_avg_sq, state)
    native.call("nrm_adam_step_dev", native.ptr(param), native.ptr(grad), native.ptr(exp_avg), native.ptr(exp_avg_sq),
                param.numel(), float(lr), float(beta1), float(beta2), float(eps), float(weight_decay), native.ptr(state),
                1 if zero_grad else 0, native.stream_ptr())


adam_step = _op("adam_step", "(Tensor(a!) param, Tensor(b!) grad, Tensor(c!) exp_avg, Tensor(d!) exp_avg_sq, Tensor(e!) state, "
                "float lr, float beta1, float beta2, float eps, float weight_decay, bool zero_grad) -> ()",
                _adam_step_impl, lambda *a: None)

OPS = ("pwattn_fwd", "pwattn_bwd", "linear_fwd", "linear_bwd", "small_linear_relu_fwd", "small_linear_relu_bwd", "mlp_gelu_fwd", "mlp_gelu_bwd", "batch_norm_stats", "batch_norm_apply",
       "batch_norm_bwd", "gate_block_fwd", "gate_block_bwd", "concat_cols",
       "weighted_pool_fwd", "weighted_pool_bwd", "attend_pool_fwd", "attend_pool_bwd", "softmax_bce_loss", "frontend_fwd", "frontend_bwd",
       "frontend_pair_fwd", "frontend_pair_bwd", "row_auc", "adam_step")
