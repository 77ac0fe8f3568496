"""nn.Module mirror of the reference model interface.

Same class names, constructor arguments, forward signatures, reachable attributes and
``state_dict`` keys as the reference's ``models/`` package, so its checkpoints load and its
train/test drivers can call these classes unchanged:

  MLP(input_dim, output_dim, activation_type='gelu')                 models/attention_model.py:10-32
  PointwiseAttention(input_dim)                                      models/attention_model.py:34-44
  PointwiseAttentionExpanded(input_dim)                              models/attention_model.py:47-97
  UserInvariantInterestModel(embed_setting=[32,16,8,8])              models/user_invariant_interest_model.py:10-89
  UserInstantInterestModel(output_dim)                               models/user_instant_interest_model.py:10-23
  UserModel(user_num=0)  .forward(x_history, x_target, x_global), .loss(id, out, label, alpha)
                                                                     models/user_model.py:12-43

Modules can be built, pickled and (de)serialised on the CPU; ``forward`` needs an MI355X: the hot
ops go through the C-ABI HIP kernels (ops.py) and raise on CPU tensors.
"""
from __future__ import annotations

import os

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops
from .config import Dims, TIME_TABLE_ROWS, model_config

_ACTIVATIONS = {
    "relu": nn.ReLU, "gelu": nn.GELU, "tanh": nn.Tanh, "sigmoid": nn.Sigmoid,
    "leaky_relu": nn.LeakyReLU, "elu": nn.ELU,
}


class MLP(nn.Module):
    """Linear(d, d//4) -> activation -> Linear(d//4, out); unknown activation names mean GELU."""

    def __init__(self, input_dim, output_dim, activation_type="gelu"):
        super().__init__()
        self.fc1 = nn.Linear(input_dim, input_dim // 4)
        self.fc2 = nn.Linear(input_dim // 4, output_dim)
        self.activation_type = activation_type
        self.activation = _ACTIVATIONS.get(str(activation_type).lower(), nn.GELU)()

    def forward(self, x):
        return self.forward_times(x, None)

    def forward_times(self, x, mul):
        """forward(x) [* mul]: the product (the gate of models/user_model.py:33) rides in fc2's GEMM epilogue."""
        ops._require_gpu(x)
        if isinstance(self.activation, nn.GELU) and self.activation.approximate == "none":
            # one autograd node: bias + exact GELU fused into fc1's GEMM, GELU' fused into the backward GEMM of fc2
            return ops.mlp_gelu(x, self.fc1.weight, self.fc1.bias, self.fc2.weight, self.fc2.bias, mul=mul)
        hidden = self.activation(ops.linear(x, self.fc1.weight, self.fc1.bias))
        y = ops.linear(hidden, self.fc2.weight, self.fc2.bias)
        return y if mul is None else y * mul


def _scores(mlp: MLP, target, history, mma=None):
    if not isinstance(mlp.activation, nn.GELU):
        raise RuntimeError("the HIP attention kernel implements the reference default (exact GELU) only")
    return ops.pointwise_attention_scores(target, history, mlp.fc1.weight, mlp.fc1.bias,
                                          mlp.fc2.weight, mlp.fc2.bias, mma=mma)


class PointwiseAttentionExpanded(nn.Module):
    """score[b,t,h] = MLP(cat[h, t, t-h, t*h]) for every (candidate, history) pair -> [B,T,H,1]."""

    mma = None          # arithmetic of the contraction: None = ops.set_attention_arithmetic default, 'f32' or 'bf16'

    def __init__(self, input_dim):
        super().__init__()
        self.mlp = MLP(input_dim * 4, 1)

    def forward(self, target, history):
        if target.dim() == 2:                      # a single target per impression
            target = target.unsqueeze(1)
        return _scores(self.mlp, target, history, self.mma).unsqueeze(-1)


class PointwiseAttention(nn.Module):
    """Non-broadcast variant (unused by UserModel): target and history have the same [N, D] shape,
    one score per row.  Runs the same kernel with T = H = 1."""

    def __init__(self, input_dim):
        super().__init__()
        self.mlp = MLP(input_dim * 4, 1)

    def forward(self, target, history):
        lead = target.shape[:-1]
        d = target.shape[-1]
        history = history.expand_as(target)
        s = _scores(self.mlp, target.reshape(-1, 1, d), history.reshape(-1, 1, d))
        return s.reshape(*lead, 1)


class UserInvariantInterestModel(nn.Module):
    """Embeds the packed history / candidate rows and pools the history with two pointwise
    attentions (label features, text+image vector) -> (eu_H, ec)."""

    def __init__(self, embed_setting=[32, 16, 8, 8]):
        super().__init__()
        self.embed_setting = embed_setting
        dims = Dims.from_config(embed_setting, model_config)
        self._dims = dims
        e0, e1, e2, e3 = dims.embed_setting
        # column widths of a packed row: time4 | text_img | category | sub-categories | sentiment | type | read | scroll
        self.slice_len_list = [4, dims.pca_vector, 1, dims.n_subcat, dims.n_sentiment, 1, 1, 1]
        self.category_embedding = nn.Sequential(nn.Embedding(dims.category_label_num, e0))
        self.sentiment_embedding = nn.Sequential(nn.Linear(dims.n_sentiment, e1), nn.ReLU())
        self.type_embedding = nn.Sequential(nn.Embedding(dims.n_type, e2))
        self.w1 = nn.Linear(dims.label_dim + 2, dims.label_dim)
        self.year_embedding = nn.Sequential(nn.Embedding(TIME_TABLE_ROWS[0], e3))
        self.month_embedding = nn.Sequential(nn.Embedding(TIME_TABLE_ROWS[1], e3))
        self.day_embedding = nn.Sequential(nn.Embedding(TIME_TABLE_ROWS[2], e3))
        self.hour_embedding = nn.Sequential(nn.Embedding(TIME_TABLE_ROWS[3], e3))
        self.label_attention = PointwiseAttentionExpanded(dims.label_dim)
        self.text_img_attention = PointwiseAttentionExpanded(dims.pca_vector)

    # -- helpers ---------------------------------------------------------------------------
    def slice_x(self, x, n):
        return list(torch.split(x[:, :, :sum(self.slice_len_list[:n])], self.slice_len_list[:n], dim=2))

    def feature_embedding(self, category, sub_category, sentiment, type):
        table = self.category_embedding[0].weight
        cat = F.embedding(category[..., 0].long(), table)
        sub = F.embedding(sub_category.long(), table).mean(dim=2)       # mean includes padding id 0
        sen = self.sentiment_embedding(sentiment)
        typ = F.embedding(type[..., 0].long(), self.type_embedding[0].weight)
        return torch.cat((cat + sub, sen, typ), dim=2)

    def time_embedding(self, time):
        idx = time.long()
        return (self.year_embedding[0](idx[..., 0]) + self.month_embedding[0](idx[..., 1])
                + self.day_embedding[0](idx[..., 2]) + self.hour_embedding[0](idx[..., 3]))

    def _embed(self, x, behaviour):
        """Fused front end (slice_x + feature_embedding + time_embedding [+ read_time, scroll]) on packed rows."""
        sen = self.sentiment_embedding[0]
        return ops.frontend(x, behaviour, self._dims.n_subcat, self._dims.pca_vector,
                            self.category_embedding[0].weight, sen.weight, sen.bias, self.type_embedding[0].weight,
                            self.year_embedding[0].weight, self.month_embedding[0].weight,
                            self.day_embedding[0].weight, self.hour_embedding[0].weight)

    def forward(self, x_history, x_target):
        pooled_lab, pooled_ti, lab_t, ti_t = self.forward_parts(x_history, x_target)
        return ops.concat_last((pooled_lab, pooled_ti)), ops.concat_last((lab_t, ti_t))       # (eu_H, ec): :81,:88

    def forward_parts(self, x_history, x_target):
        """The four column blocks of forward()'s two results -- eu_H = [pooled_lab | pooled_ti], ec = [lab_t | ti_t] -- before their
        concatenation: UserModel.forward writes them into the head's [eu_H | eu_L | ec] rows with ONE launch instead of three
        (two here, one there: 0.14 ms of a 31 ms step at C3, 50 us of 3.2 ms at C2)."""
        ops._require_gpu(x_history, x_target)
        if x_history.shape[0] * x_history.shape[1] == 0 or x_target.shape[1] == 0:
            # same error as the reference, whose feature_embedding reshapes with a -1 dimension (:59); the ops below
            # would accept empty inputs (ops._degenerate), the drop-in keeps the reference's behaviour
            raise RuntimeError("cannot reshape tensor of 0 elements (empty batch / history / candidate list)")
        # [B,H,D_l+2], [B,H,P], [B,T,D_l], [B,T,P]: both row sets through one autograd node (their table gradients share one arena)
        sen = self.sentiment_embedding[0]
        lab_h, ti_h, lab_t, ti_t = ops.frontend_pair(
            x_history, x_target, self._dims.n_subcat, self._dims.pca_vector, self.category_embedding[0].weight, sen.weight, sen.bias,
            self.type_embedding[0].weight, self.year_embedding[0].weight, self.month_embedding[0].weight,
            self.day_embedding[0].weight, self.hour_embedding[0].weight)
        lab_h = ops.linear(lab_h, self.w1.weight, self.w1.bias)

        # The two attentions (label features / text+image vector) share no data until the concat, so the second one is issued on
        # a side stream and overlaps the first -- in eager mode and as two parallel branches of the captured HIP graph; autograd
        # runs every backward node on the stream of its forward and joins the streams at the end of backward().  On small
        # shapes (C1, C2, the reference's default sizes) neither attention fills the chip; on full-size batches the HBM-bound
        # and the MFMA-bound kernels of the two branches overlap (ops.BRANCH_STREAMS_MAX_ELEMS).
        side = ops.branch_stream(lab_t) if self._two_streams(lab_t, lab_h) else None
        if side is not None:
            main = torch.cuda.current_stream()
            side.wait_stream(main)
            ti_t.record_stream(side)            # allocated on the main stream, read by side-stream kernels (forward and backward)
            ti_h.record_stream(side)
            with torch.cuda.stream(side):
                pooled_ti = self._attend_and_pool(self.text_img_attention, ti_t, ti_h)
            pooled_lab = self._attend_and_pool(self.label_attention, lab_t, lab_h)
            main.wait_stream(side)
            pooled_ti.record_stream(main)
        else:
            pooled_lab = self._attend_and_pool(self.label_attention, lab_t, lab_h)
            pooled_ti = self._attend_and_pool(self.text_img_attention, ti_t, ti_h)
        return pooled_lab, pooled_ti, lab_t, ti_t

    @staticmethod
    def _attend_and_pool(attention, t, h):
        # un-normalised weighted pool: sum_h score * history  (no softmax, padding not masked)
        mlp = getattr(attention, "mlp", None)
        if type(attention) is PointwiseAttentionExpanded and isinstance(mlp, MLP) and isinstance(mlp.activation, nn.GELU) \
                and mlp.activation.approximate == "none" and t.dim() == 3:
            # scores + pool as one autograd node (the backward chains their kernels: ops._attend_pool_bwd_impl)
            return ops.attend_and_pool(t, h, mlp.fc1.weight, mlp.fc1.bias, mlp.fc2.weight, mlp.fc2.bias, mma=attention.mma)
        s = attention(t, h)                                            # [B,T,H,1]
        return ops.weighted_pool(s.squeeze(-1), h)

    two_streams = None          # None: by size; True / False force it (env NRM_BRANCH_STREAMS=0|1 overrides both)

    def _two_streams(self, t, h):
        return self.uses_two_streams(t.shape[0] * t.shape[1] * h.shape[1] * h.shape[2])

    def uses_two_streams(self, z_elems):
        """Whether forward() issues its second attention on the side stream for attentions of ``z_elems`` = B*T*H*D."""
        forced = os.environ.get("NRM_BRANCH_STREAMS")
        if forced is not None:
            return forced == "1"
        if self.two_streams is not None:
            return bool(self.two_streams)
        return z_elems <= ops.BRANCH_STREAMS_MAX_ELEMS


class UserInstantInterestModel(nn.Module):
    """ReLU(Linear(3 -> output_dim)) on the per-candidate popularity scalars."""

    def __init__(self, output_dim):
        super().__init__()
        self.output_dim = output_dim
        self.out_fc = nn.Sequential(nn.Linear(3, output_dim), nn.ReLU())

    def forward(self, x_global):
        ops._require_gpu(x_global)
        fc = self.out_fc[0]
        # 3 -> 8 columns: one thread per row forward, a register reduction over the rows backward (ops.small_linear_relu; reads
        # the DataLoader's float64 rows directly)
        return ops.small_linear_relu(x_global, fc.weight, fc.bias)


class UserModel(nn.Module):
    """cat[eu_H, eu_L, ec] -> BatchNorm gate -> MLP -> MLP -> one logit per candidate."""

    def __init__(self, user_num=0):
        super().__init__()
        self.invariant_interest_model = UserInvariantInterestModel()
        self.instant_interest_model = UserInstantInterestModel(8)
        width = (sum(self.invariant_interest_model.embed_setting) + model_config["pca_vector"]) * 2 \
            + self.instant_interest_model.output_dim
        self.bn = nn.BatchNorm1d(width)
        self.gate = MLP(width, width)
        self.mlp = MLP(width, width)
        self.out_mlp = MLP(width, 1)
        self.delta = nn.Parameter(torch.zeros(user_num + 1))
        self.bce_loss = nn.BCELoss()
        self.softmax = nn.Softmax(dim=1)

    def forward(self, x_history, x_target, x_global):
        inv = self.invariant_interest_model
        if type(inv) is UserInvariantInterestModel and not (inv._forward_hooks or inv._forward_pre_hooks):
            # e = cat[eu_H, eu_L, ec] (:31) with eu_H and ec never materialised on their own: one concat launch of five blocks
            pooled_lab, pooled_ti, lab_t, ti_t = inv.forward_parts(x_history, x_target)
            eu_L = self.instant_interest_model(x_global)
            e = ops.concat_last((pooled_lab, pooled_ti, eu_L, lab_t, ti_t))
        else:                                   # a hooked / substituted sub-model is called as the reference calls it (:28)
            eu_H, ec = inv(x_history, x_target)
            eu_L = self.instant_interest_model(x_global)
            e = ops.concat_last((eu_H, eu_L, ec))
        B, T, N = e.shape
        rows = e.reshape(B * T, N)
        g = self.gate
        if isinstance(g.activation, nn.GELU) and g.activation.approximate == "none":
            # BatchNorm -> gate MLP -> product with the RAW concat as one autograd node
            gated = ops.gate_block(rows, self.bn, g.fc1.weight, g.fc1.bias, g.fc2.weight, g.fc2.bias)
        else:
            gated = g.forward_times(ops.batch_norm(rows, self.bn), rows)     # the gate multiplies the RAW concat
        return self.out_mlp(self.mlp(gated)).reshape(B, T)

    def loss(self, id, out, label, alpha=0.95):
        # any number of candidates: one wave per impression, in registers up to T = 256, re-reading the row beyond (csrc/pool_loss.hip)
        return ops.softmax_bce_loss(out, self.delta, label, id, alpha)
