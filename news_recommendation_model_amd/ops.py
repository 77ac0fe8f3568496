"""Autograd-visible operators over the C-ABI kernels (include/nrm_hotpath.h).

``pointwise_attention_scores`` replaces the body of the reference
``PointwiseAttentionExpanded.forward`` (models/attention_model.py:52-97).  Inputs must live on an
MI355X; anything else raises -- there is no CPU path in the product.
"""
from __future__ import annotations

import torch

from . import native


def _require_gpu(*tensors):
    for t in tensors:
        if not t.is_cuda:
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


class _PointwiseAttentionScores(torch.autograd.Function):
    """s[b,t,h] = fc2(GELU(fc1(cat[h, t, t-h, t*h]))) with fc1 = [W_h|W_t|W_d|W_p] re-associated as
    z = h(W_h-W_d)^T + b1 + t(W_t+W_d)^T + sum_d W_p[:,d] t_d h_d  (SURVEY.md §8 a7)."""

    @staticmethod
    def forward(ctx, t, h, w1, b1, w2, b2):
        _require_gpu(t, h, w1, b1, w2, b2)
        B, T, D = t.shape
        H = h.shape[1]
        if h.shape[0] != B or h.shape[2] != D or tuple(w1.shape) != (D, 4 * D):
            raise RuntimeError(f"pointwise attention: target {tuple(t.shape)}, history {tuple(h.shape)}, "
                               f"fc1 {tuple(w1.shape)} do not agree")
        t, h, w1, b1 = _f32c(t), _f32c(h), _f32c(w1), _f32c(b1)
        w2v, b2 = _f32c(w2).reshape(-1), _f32c(b2).reshape(-1)
        w_h, w_t, w_d = w1[:, :D], w1[:, D:2 * D], w1[:, 2 * D:3 * D]
        a_h = w_h - w_d
        a_t = w_t + w_d
        u, _ = _gemm_nt(h.reshape(B * H, D), a_h.contiguous(), D, 1, D, D, b1, 0)      # [B*H, D]
        v, _ = _gemm_nt(t.reshape(B * T, D), a_t.contiguous(), D, 1, D, D, None, 0)    # [B*T, D]
        need_grad = any(ctx.needs_input_grad) and torch.is_grad_enabled()      # no [B,T,H,D] buffer under no_grad / eval
        st = native.stream_ptr()
        packed = torch.empty(native.load().nrm_pwattn_packed_floats(D), dtype=torch.float32, device=t.device)
        native.call("nrm_pwattn_pack_wp", native.ptr(w1), 4 * D, D, native.ptr(packed), st)
        s = torch.empty(B, T, H, dtype=torch.float32, device=t.device)
        z = torch.empty(B, T, H, D, dtype=torch.float32, device=t.device) if need_grad else None
        native.call("nrm_pwattn_fwd", native.ptr(t), native.ptr(h), native.ptr(u), native.ptr(v),
                    native.ptr(packed), native.ptr(w2v), native.ptr(b2),
                    native.ptr(z) if z is not None else None, native.ptr(s), B, T, H, D, st)
        if need_grad:
            ctx.save_for_backward(t, h, w1, w2v, z)
            ctx.w2_shape = tuple(w2.shape)
            ctx.z_consumed = False
        return s

    @staticmethod
    def backward(ctx, ds):
        if ctx.z_consumed:
            # the saved pre-activation is overwritten in place by dz below (2.46 GB at C3: no second copy), so the
            # graph can be walked once; the reference's autograd would allow retain_graph=True here
            raise RuntimeError("pointwise attention: backward through this graph a second time is not supported "
                               "(the saved pre-activation buffer was consumed by the first backward); "
                               "re-run the forward instead of retain_graph=True")
        ctx.z_consumed = True
        t, h, w1, w2v, z = ctx.saved_tensors
        B, T, D = t.shape
        H = h.shape[1]
        st = native.stream_ptr()
        ds = _f32c(ds)
        dw2 = torch.zeros(D, dtype=torch.float32, device=t.device)
        # one pass over z: z -> dz in place (the saved tensor is consumed: a second backward through it is
        # not supported), du = sum_t dz, dv = sum_h dz, dw2
        du = torch.empty(B, H, D, dtype=torch.float32, device=t.device)
        dv = torch.empty(B, T, D, dtype=torch.float32, device=t.device)
        native.call("nrm_pwattn_bwd_dz", native.ptr(z), native.ptr(ds), native.ptr(w2v), native.ptr(dw2),
                    native.ptr(du), native.ptr(dv), B, T, H, D, st)
        dz = z
        db2 = ds.sum().reshape(1)
        w_h, w_t, w_d = w1[:, :D], w1[:, D:2 * D], w1[:, 2 * D:3 * D]
        a_h = w_h - w_d
        a_t = w_t + w_d
        du2, dv2 = du.reshape(B * H, D), dv.reshape(B * T, D)
        da_h, db1 = _gemm_tn(du2, h.reshape(B * H, D), True)      # du^T h, and db1 = column sums of du
        da_t, _ = _gemm_tn(dv2, t.reshape(B * T, D), False)
        a_hc, a_tc = a_h.contiguous(), a_t.contiguous()
        dh = _gemm_nt(du2, a_hc, 1, D, D, D, None, 0)[0].reshape(B, H, D).contiguous()     # du A_h
        dt = _gemm_nt(dv2, a_tc, 1, D, D, D, None, 0)[0].reshape(B, T, D).contiguous()     # dv A_t
        nsplit = native.load().nrm_pwattn_bwd_nsplit(B, T, H, D)
        ws = torch.empty(nsplit, D, D, dtype=torch.float32, device=t.device)
        wp = w1[:, 3 * D:]                                   # view, row stride 4D
        # two launches: (b,t)-grouped -> dt + dW_p slabs, (b,h)-grouped -> dh (issued separately so that
        # bench.py can time each kernel with its own event pair)
        for passes, tag in ((1, "pwattn_bwd_e_bt"), (2, "pwattn_bwd_e_bh")):
            native.call("nrm_pwattn_bwd_contract", native.ptr(dz), native.ptr(t), native.ptr(h),
                        native.ptr(wp), 4 * D, native.ptr(dt), native.ptr(dh), native.ptr(ws), B, T, H, D,
                        passes, st, tag=tag)
        dwp = ws.sum(dim=0).t()                             # slabs hold dW_p^T ([d][k])
        dw1 = torch.cat([da_h, da_t, da_t - da_h, dwp], dim=1)
        return dt, dh, dw1, db1, dw2.reshape(ctx.w2_shape), db2


def pointwise_attention_scores(target, history, fc1_weight, fc1_bias, fc2_weight, fc2_bias):
    """[B,T,D] x [B,H,D] -> [B,T,H] scores (fp32).

    The kernels need the feature width to be a multiple of 4 (float4 rows).  Any other D is zero-padded here with
    differentiable ops: padded features contribute exactly 0 to every term (their fc1 rows/columns and fc2 weights
    are 0, gelu(0) = 0), and autograd slices the gradients back."""
    D = target.shape[-1]
    if target.shape[0] * target.shape[1] * history.shape[1] == 0:
        return _degenerate((target.shape[0], target.shape[1], history.shape[1]), target, history, fc1_weight, fc1_bias,
                           fc2_weight, fc2_bias)
    if D % 4 == 0:
        return _PointwiseAttentionScores.apply(target, history, fc1_weight, fc1_bias, fc2_weight, fc2_bias)
    _require_gpu(target, history, fc1_weight)
    P = _pad4(D) - D
    pad = torch.nn.functional.pad
    blocks = [pad(fc1_weight[:, i * D:(i + 1) * D], (0, P, 0, P)) for i in range(4)]     # [D4, D4] each
    return _PointwiseAttentionScores.apply(pad(target.to(torch.float32), (0, P)), pad(history.to(torch.float32), (0, P)),
                                           torch.cat(blocks, dim=1), pad(fc1_bias, (0, P)),
                                           pad(fc2_weight.reshape(1, D), (0, P)), fc2_bias)


# ------------------------------------------------------------------------------------------------ dense layers
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


def _gemm_nt(x, w_src, row_stride, col_stride, n_out, k_red, bias, epilogue, z=None):
    """y[M, n_out] = epilogue(x[M, k_red] * Wlogical^T), Wlogical[r, c] = w_src[r*row_stride + c*col_stride]."""
    lib = native.load()
    st = native.stream_ptr()
    M = x.shape[0]
    packed = torch.empty(lib.nrm_gemm_packed_floats(n_out, k_red), dtype=torch.float32, device=x.device)
    native.call("nrm_gemm_pack", native.ptr(w_src), row_stride, col_stride, n_out, k_red, native.ptr(packed), st)
    ldy = _pad4(n_out)
    y = torch.empty(M, ldy, dtype=torch.float32, device=x.device)
    if epilogue == 1:
        z = torch.empty(M, ldy, dtype=torch.float32, device=x.device)
    native.call("nrm_gemm_nt", native.ptr(x), x.stride(0), M, native.ptr(packed), n_out, k_red,
                native.ptr(bias) if bias is not None else None, native.ptr(y), ldy,
                native.ptr(z) if z is not None else None, z.stride(0) if z is not None else 0, epilogue, st)
    return y[:, :n_out], (z[:, :n_out] if epilogue == 1 else None)


def _gemm_tn(a, b, want_colsum):
    """(sum_r a[r,i] b[r,j]) as [ni, nj], and optionally sum_r a[r,i]."""
    lib = native.load()
    st = native.stream_ptr()
    R, ni = a.shape
    nj = b.shape[1]
    nsplit = lib.nrm_gemm_tn_nsplit(ni, nj, R)
    ldws = _pad4(ni)
    ws = torch.empty(nsplit, nj, ldws, dtype=torch.float32, device=a.device)
    cs = torch.empty(nsplit, ldws, dtype=torch.float32, device=a.device) if want_colsum else None
    native.call("nrm_gemm_tn", native.ptr(a), a.stride(0), ni, native.ptr(b), b.stride(0), nj, R,
                native.ptr(ws), ldws, native.ptr(cs) if cs is not None else None, st)
    c = ws.sum(dim=0)[:, :ni].t()
    return c, (cs.sum(dim=0)[:ni] if want_colsum else None)


class _Linear(torch.autograd.Function):
    """y = act(x W^T + b) for a 2-D x; act in {none, exact GELU}.  Forward and both gradients run on the
    fp32 MFMA GEMM kernels (csrc/gemm.hip)."""

    @staticmethod
    def forward(ctx, x, weight, bias, gelu):
        _require_gpu(x, weight)
        x = _rows(x)
        w = _f32c(weight)
        N, K = w.shape
        if x.shape[1] != K:
            raise RuntimeError(f"linear: input has {x.shape[1]} features, weight expects {K}")
        b = _f32c(bias) if bias is not None else None
        y, z = _gemm_nt(x, w, K, 1, N, K, b, 1 if gelu else 0)
        ctx.save_for_backward(x, w, z)
        ctx.has_bias = bias is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w, z = ctx.saved_tensors
        N, K = w.shape
        if z is not None:
            dy = torch.ops.aten.gelu_backward(dy, z)            # exact-erf GELU' (elementwise)
        dy = _rows(dy)
        dx = dw = db = None
        if ctx.needs_input_grad[1] or (ctx.has_bias and ctx.needs_input_grad[2]):
            dw, db = _gemm_tn(dy, x, ctx.has_bias)              # dW[n,k] = sum_m dy[m,n] x[m,k]
        if ctx.needs_input_grad[0]:
            dx, _ = _gemm_nt(dy, w, 1, K, K, N, None, 0)        # dX = dY W : rows of the packed operand = k
        return dx, dw, (db if ctx.has_bias else None), None


def linear(x, weight, bias=None, gelu=False):
    """nn.Linear (optionally followed by exact GELU) on the last dimension of x."""
    lead = x.shape[:-1]
    if x.numel() == 0 and x.shape[-1] == weight.shape[1]:
        return _degenerate((*lead, weight.shape[0]), x, weight, bias)
    y = _Linear.apply(x.reshape(-1, x.shape[-1]), weight, bias, bool(gelu))
    return y.reshape(*lead, weight.shape[0])


class _MlpGelu(torch.autograd.Function):
    """fc2(gelu(fc1(x))) for a 2-D x (reference models/attention_model.py:29-32 with the default activation) as ONE
    autograd node: forward = two GEMMs (bias + exact GELU fused into the first, its pre-activation saved); backward =
    dW2/db2, then d(pre-activation) = (dY W2) * gelu'(z) with the GELU derivative fused into that GEMM's epilogue
    (NRM_EPI_DGELU), then dW1/db1 and dX -- no elementwise pass over the hidden activations in either direction."""

    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2):
        _require_gpu(x, w1, w2)
        x = _rows(x)
        w1, w2 = _f32c(w1), _f32c(w2)
        N1, K1 = w1.shape
        N2, K2 = w2.shape
        if x.shape[1] != K1 or K2 != N1:
            raise RuntimeError(f"mlp: input has {x.shape[1]} features, fc1 expects {K1}, fc2 expects {K2} hidden")
        hidden, z = _gemm_nt(x, w1, K1, 1, N1, K1, _f32c(b1) if b1 is not None else None, 1)
        y, _ = _gemm_nt(hidden, w2, K2, 1, N2, K2, _f32c(b2) if b2 is not None else None, 0)
        ctx.save_for_backward(x, w1, w2, hidden, z)
        ctx.has_bias = (b1 is not None, b2 is not None)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w1, w2, hidden, z = ctx.saved_tensors
        N1, K1 = w1.shape
        N2, K2 = w2.shape
        dy = _rows(dy)
        need = ctx.needs_input_grad
        dw2 = db2 = dw1 = db1 = dx = None
        if need[3] or (ctx.has_bias[1] and need[4]):
            dw2, db2 = _gemm_tn(dy, hidden, ctx.has_bias[1])               # dW2[n,k] = sum_m dy[m,n] hidden[m,k]
        # d(pre-activation of fc1) = (dY W2) * gelu'(z): epilogue 2 reads z and writes the product
        dz, _ = _gemm_nt(dy, w2, 1, K2, K2, N2, None, 2, z=z)
        if need[1] or (ctx.has_bias[0] and need[2]):
            dw1, db1 = _gemm_tn(dz, x, ctx.has_bias[0])
        if need[0]:
            dx, _ = _gemm_nt(dz, w1, 1, K1, K1, N1, None, 0)
        return dx, dw1, (db1 if ctx.has_bias[0] else None), dw2, (db2 if ctx.has_bias[1] else None)


def mlp_gelu(x, fc1_weight, fc1_bias, fc2_weight, fc2_bias):
    """Linear -> exact GELU -> Linear on the last dimension of x (one fused autograd node)."""
    lead = x.shape[:-1]
    if x.numel() == 0 and x.shape[-1] == fc1_weight.shape[1]:
        return _degenerate((*lead, fc2_weight.shape[0]), x, fc1_weight, fc1_bias, fc2_weight, fc2_bias)
    y = _MlpGelu.apply(x.reshape(-1, x.shape[-1]), fc1_weight, fc1_bias, fc2_weight, fc2_bias)
    return y.reshape(*lead, fc2_weight.shape[0])


# ------------------------------------------------------------------------------------------------ BatchNorm1d
class _BatchNorm(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, running_mean, running_var, training, momentum, eps):
        _require_gpu(x, weight, bias)
        x = _rows(x)
        R, N = x.shape
        ld = x.stride(0)
        st = native.stream_ptr()
        w, b = _f32c(weight), _f32c(bias)
        if training:
            s = torch.zeros(2, N, dtype=torch.float32, device=x.device)
            native.call("nrm_colreduce", 0, native.ptr(x), None, None, None, native.ptr(s[0]), None, R, N, ld, st)
            mean = s[0] / R
            native.call("nrm_colreduce", 1, native.ptr(x), None, native.ptr(mean), None, native.ptr(s[1]), None, R, N, ld, st)
            var = s[1] / R
            with torch.no_grad():
                running_mean.mul_(1 - momentum).add_(mean, alpha=momentum)
                running_var.mul_(1 - momentum).add_(var, alpha=momentum * R / max(R - 1, 1))
        else:
            mean, var = _f32c(running_mean), _f32c(running_var)
        rstd = torch.rsqrt(var + eps)
        y = torch.empty(R, ld, dtype=torch.float32, device=x.device)
        native.call("nrm_bn_apply", native.ptr(x), native.ptr(mean), native.ptr(rstd), native.ptr(w), native.ptr(b),
                    native.ptr(y), R, N, ld, st)
        ctx.save_for_backward(x, mean, rstd, w)
        ctx.training = training
        return y[:, :N]

    @staticmethod
    def backward(ctx, dy):
        x, mean, rstd, w = ctx.saved_tensors
        R, N = x.shape
        st = native.stream_ptr()
        if not (dy.dim() == 2 and dy.stride(1) == 1 and dy.stride(0) == x.stride(0) and dy.dtype == torch.float32
                and dy.data_ptr() % 16 == 0):
            buf = torch.zeros(R, x.stride(0), dtype=torch.float32, device=x.device)
            buf[:, :N] = dy
            dy = buf[:, :N]
        ld = x.stride(0)
        s = torch.zeros(2, N, dtype=torch.float32, device=x.device)
        native.call("nrm_colreduce", 2, native.ptr(x), native.ptr(dy), native.ptr(mean), native.ptr(rstd),
                    native.ptr(s[0]), native.ptr(s[1]), R, N, ld, st)
        dx = torch.empty(R, ld, dtype=torch.float32, device=x.device)
        native.call("nrm_bn_backward", native.ptr(x), native.ptr(dy), native.ptr(mean), native.ptr(rstd), native.ptr(w),
                    native.ptr(s[0]), native.ptr(s[1]), native.ptr(dx), R, N, ld, 1 if ctx.training else 0, st)
        return dx[:, :N], s[1], s[0], None, None, None, None, None


def batch_norm(x, bn):
    """nn.BatchNorm1d(x) for a 2-D x through the HIP column kernels.  Non-default module configurations
    (no affine / no running stats / cumulative momentum) and widths that are not a multiple of 4 use the
    module itself."""
    if (x.dim() != 2 or x.shape[1] % 4 or not bn.affine or not bn.track_running_stats or bn.momentum is None):
        return bn(x)
    training = bn.training
    if x.shape[0] <= (1 if training else 0):
        return bn(x)          # 0/1 rows in training: nn.BatchNorm1d raises its ValueError; 0 rows in eval: empty result
    if training:
        bn.num_batches_tracked.add_(1)
    return _BatchNorm.apply(x, bn.weight, bn.bias, bn.running_mean, bn.running_var, training, float(bn.momentum),
                            float(bn.eps))


# ------------------------------------------------------------------------------------------------ pool
class _WeightedPool(torch.autograd.Function):
    """pooled[b,t,:] = sum_h s[b,t,h] * h[b,h,:]  (un-normalised, unmasked: reference
    models/user_invariant_interest_model.py:86-87)."""

    @staticmethod
    def forward(ctx, s, h):
        _require_gpu(s, h)
        s, h = _f32c(s), _f32c(h)
        B, T, H = s.shape
        D = h.shape[2]
        out = torch.empty(B, T, D, dtype=torch.float32, device=h.device)
        native.call("nrm_pool_bmm", native.ptr(s), T * H, H, 1, native.ptr(h), native.ptr(out), B, T, H, D, 0,
                    native.stream_ptr())
        ctx.save_for_backward(s, h)
        return out

    @staticmethod
    def backward(ctx, g):
        s, h = ctx.saved_tensors
        B, T, H = s.shape
        D = h.shape[2]
        g = _f32c(g)
        st = native.stream_ptr()
        ds = torch.empty(B, T, H, dtype=torch.float32, device=h.device)
        native.call("nrm_pool_rowdot", native.ptr(g), native.ptr(h), native.ptr(ds), B, T, H, D, st)
        dh = torch.empty(B, H, D, dtype=torch.float32, device=h.device)
        native.call("nrm_pool_bmm", native.ptr(s), T * H, 1, H, native.ptr(g), native.ptr(dh), B, H, T, D, 0, st)
        return ds, dh


def weighted_pool(scores, history):
    D = history.shape[-1]
    if scores.numel() == 0:                                  # empty batch / no candidates / empty history: sum of nothing
        return _degenerate((scores.shape[0], scores.shape[1], D), scores, history)
    if D % 4 == 0:
        return _WeightedPool.apply(scores, history)
    _require_gpu(scores, history)
    return _WeightedPool.apply(scores, torch.nn.functional.pad(history.to(torch.float32), (0, _pad4(D) - D)))[..., :D]


# ------------------------------------------------------------------------------------------------ loss
class _SoftmaxBceLoss(torch.autograd.Function):
    """The two-term BCE-on-softmax loss of reference models/user_model.py:37-43, value and gradients in one
    kernel (one wave per impression)."""

    @staticmethod
    def forward(ctx, out, delta, label, user_id, alpha):
        _require_gpu(out, delta, label, user_id)
        B, T = out.shape
        o = _f32c(out)
        y = _f32c(label)
        uid = user_id.to(torch.int64).contiguous()
        d = _f32c(delta)
        loss = torch.zeros(1, dtype=torch.float32, device=out.device)
        dout = torch.empty(B, T, dtype=torch.float32, device=out.device)
        ddelta = torch.zeros_like(d)
        native.call("nrm_loss_fwd_bwd", native.ptr(o), native.ptr(y), native.ptr(uid), native.ptr(d), d.numel(), float(alpha),
                    B, T, native.ptr(loss), native.ptr(dout), native.ptr(ddelta), native.ptr(index_error_flag(out.device)),
                    native.stream_ptr())
        ctx.save_for_backward(dout, ddelta)
        return loss.reshape(())

    @staticmethod
    def backward(ctx, gl):
        dout, ddelta = ctx.saved_tensors
        return dout * gl, ddelta * gl, None, None, None


def softmax_bce_loss(out, delta, label, user_id, alpha):
    if out.numel() == 0:                                     # nn.BCELoss: mean over no elements
        return _degenerate((), out, delta) + float("nan")
    return _SoftmaxBceLoss.apply(out, delta, label, user_id, alpha)


# ------------------------------------------------------------------------------------------------ embedding front end
_index_error_flag = {}


def index_error_flag(device):
    """Device int32 set to 1 by the front-end kernel when a packed row holds an out-of-range table index (the
    reference raises IndexError there; the kernel clamps, flags and goes on).  Reading it synchronises, so it is
    checked by ``check_index_errors`` on request, not on every step."""
    device = torch.device(device)
    if device.index is None:
        device = torch.device(device.type, torch.cuda.current_device())
    key = str(device)
    if key not in _index_error_flag:
        _index_error_flag[key] = torch.zeros(1, dtype=torch.int32, device=device)
    return _index_error_flag[key]


def check_index_errors(device="cuda"):
    flag = index_error_flag(device)
    if int(flag.item()):
        flag.zero_()
        raise IndexError("index out of range in a packed feature row (category / type / time table)")


class _Frontend(torch.autograd.Function):
    """Packed rows [R, cols] (fp32 or fp64) -> (label rows [R, e0+e1+e2+e3(+2)], text/image rows [R, P] fp32)."""

    @staticmethod
    def forward(ctx, x, behaviour, n_sub, P, cat_tab, sen_w, sen_b, type_tab, year_tab, month_tab, day_tab, hour_tab):
        _require_gpu(x, cat_tab)
        if x.dtype not in (torch.float32, torch.float64):
            x = x.to(torch.float32)
        x = x.contiguous()
        R, xcols = x.shape
        tabs = [_f32c(t) for t in (cat_tab, sen_w, sen_b, type_tab, year_tab, month_tab, day_tab, hour_tab)]
        cat_tab, sen_w, sen_b, type_tab, year_tab, month_tab, day_tab, hour_tab = tabs
        e0, e1, e2, e3 = cat_tab.shape[1], sen_w.shape[0], type_tab.shape[1], year_tab.shape[1]
        width = e0 + e1 + e2 + e3 + (2 if behaviour else 0)
        ldlab, ldti = _pad4(width), _pad4(P)
        lab = torch.empty(R, ldlab, dtype=torch.float32, device=x.device)
        ti = torch.empty(R, ldti, dtype=torch.float32, device=x.device)
        dims = (cat_tab.shape[0], e0, e1, type_tab.shape[0], e2, year_tab.shape[0], month_tab.shape[0],
                day_tab.shape[0], hour_tab.shape[0], e3)
        native.call("nrm_frontend_fwd", native.ptr(x), 1 if x.dtype == torch.float64 else 0, R, xcols, P, n_sub,
                    1 if behaviour else 0, native.ptr(cat_tab), dims[0], e0, native.ptr(sen_w), native.ptr(sen_b), e1,
                    native.ptr(type_tab), dims[3], e2, native.ptr(year_tab), native.ptr(month_tab), native.ptr(day_tab),
                    native.ptr(hour_tab), dims[5], dims[6], dims[7], dims[8], e3,
                    native.ptr(lab), ldlab, native.ptr(ti), ldti, native.ptr(index_error_flag(x.device)),
                    native.stream_ptr())
        ctx.save_for_backward(x, sen_w, sen_b)
        ctx.geom = (behaviour, n_sub, P, dims)
        ctx.mark_non_differentiable(ti)
        return lab[:, :width], ti[:, :P]

    @staticmethod
    def backward(ctx, dlab, _dti):
        x, sen_w, sen_b = ctx.saved_tensors
        behaviour, n_sub, P, dims = ctx.geom
        n_cat, e0, e1, n_type, e2, n_year, n_month, n_day, n_hour, e3 = dims
        dlab = _rows(dlab)
        dev = x.device
        z = lambda *shape: torch.zeros(*shape, dtype=torch.float32, device=dev)
        d_cat, d_sw, d_sb, d_type = z(n_cat, e0), z(e1, 3), z(e1), z(n_type, e2)
        d_year, d_month, d_day, d_hour = z(n_year, e3), z(n_month, e3), z(n_day, e3), z(n_hour, e3)
        native.call("nrm_frontend_bwd", native.ptr(x), 1 if x.dtype == torch.float64 else 0, x.shape[0], x.shape[1], P,
                    n_sub, 1 if behaviour else 0, native.ptr(dlab), dlab.stride(0), native.ptr(sen_w), native.ptr(sen_b),
                    n_cat, e0, e1, n_type, e2, n_year, n_month, n_day, n_hour, e3,
                    native.ptr(d_cat), native.ptr(d_sw), native.ptr(d_sb), native.ptr(d_type),
                    native.ptr(d_year), native.ptr(d_month), native.ptr(d_day), native.ptr(d_hour), native.stream_ptr())
        return None, None, None, None, d_cat, d_sw, d_sb, d_type, d_year, d_month, d_day, d_hour


def frontend(x, behaviour, n_sub, P, cat_tab, sen_w, sen_b, type_tab, year_tab, month_tab, day_tab, hour_tab):
    """x [B, N, cols] -> (label rows [B, N, width], text/image rows [B, N, P])."""
    B, N = x.shape[0], x.shape[1]
    if B * N == 0:
        tabs = (cat_tab, sen_w, sen_b, type_tab, year_tab, month_tab, day_tab, hour_tab)
        width = cat_tab.shape[1] + sen_w.shape[0] + type_tab.shape[1] + year_tab.shape[1] + (2 if behaviour else 0)
        return _degenerate((B, N, width), x, *tabs), _degenerate((B, N, int(P)), x)
    lab, ti = _Frontend.apply(x.reshape(B * N, x.shape[2]), bool(behaviour), int(n_sub), int(P), cat_tab, sen_w, sen_b,
                              type_tab, year_tab, month_tab, day_tab, hour_tab)
    return lab.reshape(B, N, -1) if lab.is_contiguous() else lab.unflatten(0, (B, N)), ti.unflatten(0, (B, N))
