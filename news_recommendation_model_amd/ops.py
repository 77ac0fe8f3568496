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
        u = torch.addmm(b1, h.reshape(B * H, D), a_h.t())          # [B*H, D]
        v = t.reshape(B * T, D) @ a_t.t()                            # [B*T, D]
        need_grad = any(ctx.needs_input_grad)
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
        return s

    @staticmethod
    def backward(ctx, ds):
        t, h, w1, w2v, z = ctx.saved_tensors
        B, T, D = t.shape
        H = h.shape[1]
        st = native.stream_ptr()
        ds = _f32c(ds)
        dw2 = torch.zeros(D, dtype=torch.float32, device=t.device)
        # z -> dz in place (the saved tensor is consumed: a second backward through it is not supported)
        native.call("nrm_pwattn_bwd_dz", native.ptr(z), native.ptr(ds), native.ptr(w2v), native.ptr(dw2),
                    B * T * H, D, st)
        dz = z
        db2 = ds.sum().reshape(1)
        du = dz.sum(dim=1)                                   # [B,H,D]
        dv = dz.sum(dim=2)                                   # [B,T,D]
        db1 = du.sum(dim=(0, 1))
        w_h, w_t, w_d = w1[:, :D], w1[:, D:2 * D], w1[:, 2 * D:3 * D]
        a_h = w_h - w_d
        a_t = w_t + w_d
        du2, dv2 = du.reshape(B * H, D), dv.reshape(B * T, D)
        da_h = du2.t() @ h.reshape(B * H, D)
        da_t = dv2.t() @ t.reshape(B * T, D)
        dh = (du2 @ a_h).reshape(B, H, D)
        dt = (dv2 @ a_t).reshape(B, T, D)
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
    """[B,T,D] x [B,H,D] -> [B,T,H] scores (fp32)."""
    return _PointwiseAttentionScores.apply(target, history, fc1_weight, fc1_bias, fc2_weight, fc2_bias)
