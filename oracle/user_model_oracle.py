"""ORACLE -- test infrastructure only, never part of the product path.

A plain PyTorch-CPU fp32 restatement of the reference hot path, in the reference's own
(literal) association: the [B,T,H,4D] concat is materialised exactly as the reference
does.  It exists to (1) be pinned against the real reference in this container by
``oracle/make_golden.py`` (which also writes tests/golden/*.npz), (2) check the HIP path in
``tests/`` and ``__graft_entry__.smoke()``, and (3) be timed as the ``cpu_baseline`` leg of
``bench.py``.  Only those three may import it.  Parity status: PINNED -- every function below is
compared with the imported reference model on seeded inputs by make_golden.py (max abs
difference recorded in tests/golden/MANIFEST.json).

Parameters are a flat dict keyed by the reference's ``state_dict`` names.
All citations are relative to /root/reference.
"""
from __future__ import annotations

import math

import numpy as np
import torch
import torch.nn.functional as F


# Arithmetic of the restatement.  float32 is the reference's (models/user_invariant_interest_model.py:74-75 cast the float64
# DataLoader rows to float32) and the only mode that is pinned / timed.  ``precision(torch.float64)`` re-runs the same
# formulas in float64 for ERROR ANALYSIS in tests/ (what an fp32 sum with heavy cancellation can be held to); parameters must
# then be float64 as well (``to_torch_params(sd, dtype=torch.float64)``).
COMPUTE_DTYPE = torch.float32


class precision:
    def __init__(self, dtype):
        self.dtype = dtype

    def __enter__(self):
        global COMPUTE_DTYPE
        self.prev, COMPUTE_DTYPE = COMPUTE_DTYPE, self.dtype
        return self

    def __exit__(self, *exc):
        global COMPUTE_DTYPE
        COMPUTE_DTYPE = self.prev
        return False


# --------------------------------------------------------------------------- building blocks
def mlp(p, prefix, x):
    """Linear(d -> d//4) -> exact-erf GELU -> Linear(d//4 -> out).  models/attention_model.py:29-32
    (activation default 'gelu' = nn.GELU(), :21,27)."""
    hid = F.linear(x, p[prefix + ".fc1.weight"], p[prefix + ".fc1.bias"])
    hid = 0.5 * hid * (1.0 + torch.erf(hid * (1.0 / math.sqrt(2.0))))
    return F.linear(hid, p[prefix + ".fc2.weight"], p[prefix + ".fc2.bias"])


def pointwise_attention_scores(p, prefix, target, history):
    """score[b,t,h] = MLP(cat[h, t, t-h, t*h]) -> [B,T,H,1].  models/attention_model.py:52-97.
    A 2-D target [B,D] is treated as T=1 (:64-65)."""
    if target.dim() == 2:
        target = target[:, None, :]
    B, T, D = target.shape
    H = history.shape[1]
    te = target[:, :, None, :].expand(B, T, H, D)
    he = history[:, None, :, :].expand(B, T, H, D)
    feats = torch.cat([he, te, te - he, te * he], dim=-1)          # :81-86, order matters
    s = mlp(p, prefix + ".mlp", feats.reshape(-1, 4 * D))          # :89-92
    return s.reshape(B, T, H, 1)                                   # :95


def _split_cols(x, widths):
    out, c = [], 0
    for w in widths:
        out.append(x[:, :, c:c + w])
        c += w
    return out


def _label_features(p, category, sub_category, sentiment, typ):
    """models/user_invariant_interest_model.py:58-64.  The category table serves both the category
    and the (mean of the) sub-category ids; the mean includes padding id 0."""
    inv = "invariant_interest_model."
    table = p[inv + "category_embedding.0.weight"]
    cat = table[category[..., 0].long()]                           # [B,N,e0]
    sub = table[sub_category.long()].mean(dim=2)                   # [B,N,5,e0] -> mean over 5
    sen = torch.relu(F.linear(sentiment, p[inv + "sentiment_embedding.0.weight"],
                              p[inv + "sentiment_embedding.0.bias"]))
    typ_e = p[inv + "type_embedding.0.weight"][typ[..., 0].long()]
    return torch.cat([cat + sub, sen, typ_e], dim=2)


def _time_features(p, time4):
    """Sum of the year/month/day/hour lookups.  models/user_invariant_interest_model.py:66-71."""
    inv = "invariant_interest_model."
    out = 0
    for i, name in enumerate(("year", "month", "day", "hour")):
        out = out + p[inv + name + "_embedding.0.weight"][time4[..., i].long()]
    return out


def invariant_interest(p, x_history, x_target, n_sub=5, n_sent=3, return_aux=False):
    """(eu_H, ec).  models/user_invariant_interest_model.py:73-89."""
    inv = "invariant_interest_model."
    P = p[inv + "text_img_attention.mlp.fc2.weight"].shape[1]
    widths = [4, P, 1, n_sub, n_sent, 1, 1, 1]
    if x_history.shape[0] * x_history.shape[1] == 0 or x_target.shape[1] == 0:
        # :59 reshapes the embedded ids with a -1 dimension, which ATen refuses for 0 elements: the reference model
        # raises RuntimeError for an empty batch, an empty history and an empty candidate list (train and eval;
        # tests/golden/MANIFEST.json "degenerate").  The attention module alone accepts them (empty scores).
        raise RuntimeError("cannot reshape tensor of 0 elements (empty batch / history / candidate list)")
    time_h, ti_h, cat_h, sub_h, sen_h, typ_h, read_h, scroll_h = _split_cols(x_history.to(COMPUTE_DTYPE), widths)
    time_t, ti_t, cat_t, sub_t, sen_t, typ_t = _split_cols(x_target.to(COMPUTE_DTYPE), widths[:6])

    lab_h = torch.cat([_label_features(p, cat_h, sub_h, sen_h, typ_h), _time_features(p, time_h),
                       read_h, scroll_h], dim=2)                                   # :77
    lab_h = F.linear(lab_h, p[inv + "w1.weight"], p[inv + "w1.bias"])              # :78
    lab_t = torch.cat([_label_features(p, cat_t, sub_t, sen_t, typ_t), _time_features(p, time_t)], dim=2)  # :79
    ec = torch.cat([lab_t, ti_t], dim=2)                                           # :81

    s_lab = pointwise_attention_scores(p, inv + "label_attention", lab_t, lab_h)   # :83
    s_ti = pointwise_attention_scores(p, inv + "text_img_attention", ti_t, ti_h)   # :84
    pooled_lab = torch.sum(s_lab * lab_h[:, None], dim=2)                          # :86 (no softmax, no mask)
    pooled_ti = torch.sum(s_ti * ti_h[:, None], dim=2)                             # :87
    eu_H = torch.cat([pooled_lab, pooled_ti], dim=2)                               # :88
    if return_aux:
        return eu_H, ec, {"score_label": s_lab, "score_text_img": s_ti, "label_h": lab_h, "label_t": lab_t}
    return eu_H, ec


def instant_interest(p, x_global):
    """ReLU(Linear(3->8)).  models/user_instant_interest_model.py:20-23."""
    return torch.relu(F.linear(x_global.to(COMPUTE_DTYPE), p["instant_interest_model.out_fc.0.weight"],
                               p["instant_interest_model.out_fc.0.bias"]))


def user_model_forward(p, x_history, x_target, x_global, training=True, bn_state=None,
                       momentum=0.1, eps=1e-5, return_aux=False):
    """r[B,T].  models/user_model.py:27-35.  ``bn_state`` (dict with running_mean/var,
    num_batches_tracked) is updated in place when training, as nn.BatchNorm1d does."""
    eu_H, ec, aux = invariant_interest(p, x_history, x_target, return_aux=True)
    eu_L = instant_interest(p, x_global)
    e = torch.cat([eu_H, eu_L, ec], dim=2)                         # :31
    B, T, N = e.shape
    e2 = e.reshape(B * T, N)
    if training:
        if e2.shape[0] <= 1:      # nn.BatchNorm1d in training mode (user_model.py:32)
            raise ValueError(f"Expected more than 1 value per channel when training, got input size {list(e2.shape)}")
        mean = e2.mean(dim=0)
        var_b = e2.var(dim=0, unbiased=False)
        if bn_state is not None:
            with torch.no_grad():
                n = e2.shape[0]
                bn_state["running_mean"].mul_(1 - momentum).add_(momentum * mean)
                bn_state["running_var"].mul_(1 - momentum).add_(momentum * var_b * n / max(n - 1, 1))
                bn_state["num_batches_tracked"] += 1
    else:
        mean, var_b = p["bn.running_mean"], p["bn.running_var"]
    c = (e2 - mean) / torch.sqrt(var_b + eps) * p["bn.weight"] + p["bn.bias"]   # :32
    x = mlp(p, "gate", c) * e2                                     # :33  (gate multiplies the RAW concat)
    r = mlp(p, "out_mlp", mlp(p, "mlp", x)).reshape(B, T)          # :33-34
    if return_aux:
        aux.update({"eu_H": eu_H, "ec": ec, "eu_L": eu_L, "e2": e2, "c": c, "bn_var": var_b})
        return r, aux
    return r


def bce_mean(prob, y):
    """nn.BCELoss() (models/user_model.py:24): mean of -(y log p + (1-y) log(1-p)) with both logs clamped at
    -100, and PyTorch's guarded backward (p - y) / max(p (1 - p), 1e-12) -- a hand-written clamp(log p) has a
    0 * inf = NaN gradient where p underflows, the reference's loss does not."""
    return F.binary_cross_entropy(prob, y, reduction="mean")


def user_model_loss(p, user_id, out, label, alpha=0.95):
    """(1-alpha)*BCE(softmax_T(out)) + alpha*BCE(softmax_T(out + delta[id])).  models/user_model.py:37-43."""
    y = label.to(COMPUTE_DTYPE)
    l1 = bce_mean(torch.softmax(out, dim=1), y)
    d = p["delta"][user_id.long()][:, None].expand(-1, y.shape[1])
    l2 = bce_mean(torch.softmax(out + d, dim=1), y)
    return (1 - alpha) * l1 + alpha * l2


# --------------------------------------------------------------------------- the train.py:66-75 step
BUFFER_KEYS = ("bn.running_mean", "bn.running_var", "bn.num_batches_tracked")


def to_torch_params(sd_np, requires_grad=True, dtype=None):
    p = {}
    for k, v in sd_np.items():
        t = torch.from_numpy(np.array(v, copy=True))
        if dtype is not None and t.is_floating_point():
            t = t.to(dtype)
        if k not in BUFFER_KEYS and requires_grad:
            t.requires_grad_(True)
        p[k] = t
    return p


def adam_update(param, grad, m, v, step, lr=1e-3, b1=0.9, b2=0.999, eps=1e-8, weight_decay=1e-5):
    """torch.optim.Adam semantics used at train.py:48 (L2 folded into the gradient, bias-corrected,
    eps added after the sqrt of the corrected second moment)."""
    g = grad + weight_decay * param
    m.mul_(b1).add_(g, alpha=1 - b1)
    v.mul_(b2).addcmul_(g, g, value=1 - b2)
    bc1 = 1 - b1 ** step
    bc2 = 1 - b2 ** step
    denom = (v.sqrt() / math.sqrt(bc2)).add_(eps)
    param.addcdiv_(m, denom, value=-lr / bc1)


def train_step(p, opt_state, batch, lr=1e-3, weight_decay=1e-5, alpha=0.95):
    """forward -> loss -> backward -> Adam step -> zero_grad (train.py:69-75), train-mode BN.
    ``p`` holds leaf tensors (updated in place); ``opt_state`` = {"step": int, "m": {}, "v": {}}.
    Returns (loss, r, grads)."""
    bn_state = {k.split(".")[1]: p[k] for k in BUFFER_KEYS}
    r = user_model_forward(p, batch["x_history"], batch["x_target"], batch["x_global"],
                           training=True, bn_state=bn_state)
    loss = user_model_loss(p, batch["user_id"], r, batch["label"], alpha)
    names = [k for k in p if k not in BUFFER_KEYS]
    grads = torch.autograd.grad(loss, [p[k] for k in names], allow_unused=True)
    opt_state["step"] += 1
    gout = {}
    with torch.no_grad():
        for k, g in zip(names, grads):
            if g is None:
                g = torch.zeros_like(p[k])
            gout[k] = g
            if k not in opt_state["m"]:
                opt_state["m"][k] = torch.zeros_like(p[k])
                opt_state["v"][k] = torch.zeros_like(p[k])
            adam_update(p[k], g, opt_state["m"][k], opt_state["v"][k], opt_state["step"],
                        lr=lr, weight_decay=weight_decay)
    return loss.detach(), r.detach(), gout


# --------------------------------------------------------------------------- per-row AUC (train.py:77-80)
def row_auc(label_row, score_row):
    """Binary ROC-AUC of one impression = Mann-Whitney U with tie-averaged ranks, which is what
    sklearn.metrics.roc_auc_score (tool/evaluation.py:3-5) returns for binary labels."""
    y = np.asarray(label_row, dtype=np.float64) > 0.5
    s = np.asarray(score_row, dtype=np.float64)
    n_pos, n_neg = int(y.sum()), int((~y).sum())
    if n_pos == 0 or n_neg == 0:
        raise ValueError("Only one class present in y_true. ROC AUC score is not defined in that case.")
    pos, neg = s[y], s[~y]
    gt = (pos[:, None] > neg[None, :]).sum()
    eq = (pos[:, None] == neg[None, :]).sum()
    return float(gt + 0.5 * eq) / float(n_pos * n_neg)


def batch_auc(label, score):
    return np.array([row_auc(label[b], score[b]) for b in range(label.shape[0])])


# --------------------------------------------------------------------------- inference (test.py:31-74)
def model_test_scores(param_list, batch):
    """Restatement of reference test.py:31-74 ``model_test`` for one batch: eval-mode forward of every model,
    trailing padding common to the batch trimmed first (:48-56), softmax over candidates averaged over the models
    (:58-64), then per row either the scores as they are or -- when the row still has padding -- a second softmax
    over the de-padded slice (:66-70).  Returns a list of 1-D arrays (one per impression).
    PARITY UNPINNED for this function: test.py cannot be imported here (it imports tool.process_data, which needs
    the absent ``zstandard`` package) and the reference holds no fixture for it; it is restated from the source text."""
    xh, xt, xg = batch["x_history"], batch["x_target"], batch["x_global"]
    empty = batch["empty_num"].clone()
    trim = int(empty.min())
    if trim > 0:
        xt, xg = xt[:, :-trim], xg[:, :-trim]
        empty = empty - trim
    out = None
    with torch.no_grad():
        for p in param_list:
            pr = torch.softmax(user_model_forward(p, xh, xt, xg, training=False), dim=1)
            out = pr if out is None else out + pr
        out = out / len(param_list)
        rows = []
        for i in range(out.shape[0]):
            z = int(empty[i])
            if z > 0:
                rows.append(torch.softmax(out[i:i + 1, 0:-z], dim=1).squeeze(0).numpy())
            else:
                rows.append(out[i].numpy())
    return rows
