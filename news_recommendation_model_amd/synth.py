"""Seeded synthetic EBNeRD-shaped impressions and deterministic weights.

The row layout is the one the reference ETL emits and ``slice_x`` consumes
(reference tool/process_data.py:198,207,214-240; models/user_invariant_interest_model.py:14-22):

  history row = [year, month, day, hour | text_img P | category | sub-category x5 |
                 sentiment x3 | type | read_time | scroll]
  target  row = the same without read_time, scroll
  global  row = [inviews, pageviews, read_time] (normalised counts)

Everything here is numpy (PCG64) so the same seed gives the same bytes on every
machine; value ranges follow SURVEY.md §8d.
"""
from __future__ import annotations

import numpy as np

from .config import Dims, TIME_TABLE_ROWS


def _rows(rng, n_rows, dims: Dims, with_behaviour: bool):
    P = dims.pca_vector
    cols = dims.history_cols if with_behaviour else dims.target_cols
    x = np.zeros((n_rows, cols), dtype=np.float64)
    x[:, 0] = rng.integers(0, TIME_TABLE_ROWS[0], n_rows)
    x[:, 1] = rng.integers(0, TIME_TABLE_ROWS[1], n_rows)
    x[:, 2] = rng.integers(0, 31, n_rows)
    x[:, 3] = rng.integers(0, TIME_TABLE_ROWS[3], n_rows)
    x[:, 4:4 + P] = rng.standard_normal((n_rows, P))
    c = 4 + P
    x[:, c] = rng.integers(0, dims.category_label_num, n_rows)
    x[:, c + 1:c + 1 + dims.n_subcat] = rng.integers(0, dims.category_label_num, (n_rows, dims.n_subcat))
    c += 1 + dims.n_subcat
    which = rng.integers(0, dims.n_sentiment, n_rows)
    x[np.arange(n_rows), c + which] = rng.random(n_rows)
    c += dims.n_sentiment
    x[:, c] = rng.integers(0, dims.n_type, n_rows)
    if with_behaviour:
        x[:, c + 1] = rng.random(n_rows)
        x[:, c + 2] = rng.random(n_rows)
    return x


def make_batch(dims: Dims, B: int, H: int, T: int, seed: int = 0, user_num: int | None = None,
               pad_history: int = 0, pad_target: int = 0, dtype=np.float64):
    """One batch in the 8-field record order of reference train.py:67, minus impression_id/label_id.

    ``pad_history`` / ``pad_target`` zero the trailing rows of every impression the way the
    reference ETL pads short histories / candidate lists (all-zero rows, not masked).
    Returns a dict of numpy arrays: user_id [B] int64, x_history [B,H,*], x_target [B,T,*],
    x_global [B,T,3], label [B,T] (one-hot), empty_num [B].
    """
    rng = np.random.default_rng(seed)
    user_num = 10 * B if user_num is None else user_num
    xh = _rows(rng, B * H, dims, True).reshape(B, H, -1)
    xt = _rows(rng, B * T, dims, False).reshape(B, T, -1)
    xg = rng.random((B, T, 3)) * 1e-2
    T_live = T - pad_target
    label = np.zeros((B, T), dtype=np.float64)
    label[np.arange(B), rng.integers(0, T_live, B)] = 1.0
    uid = rng.integers(0, user_num + 1, B).astype(np.int64)
    if pad_history:
        xh[:, H - pad_history:, :] = 0.0
    if pad_target:
        xt[:, T_live:, :] = 0.0
        xg[:, T_live:, :] = 0.0
    return {
        "user_id": uid,
        "x_history": xh.astype(dtype), "x_target": xt.astype(dtype), "x_global": xg.astype(dtype),
        "label": label, "empty_num": np.full((B,), pad_target, dtype=np.int64),
        "user_num": np.int64(user_num),
    }


def state_dict_shapes(dims: Dims, user_num: int | None = None):
    """Ordered (key, shape, kind) list: the 37 state_dict entries of the reference UserModel
     plus ``delta`` when ``user_num`` is given."""
    E = dims.embed_setting
    Dl, P, N = dims.label_dim, dims.pca_vector, dims.head_dim
    inv = "invariant_interest_model."
    out = [
        (inv + "category_embedding.0.weight", (dims.category_label_num, E[0]), "emb"),
        (inv + "sentiment_embedding.0.weight", (E[1], dims.n_sentiment), "w"),
        (inv + "sentiment_embedding.0.bias", (E[1],), "b:%d" % dims.n_sentiment),
        (inv + "type_embedding.0.weight", (dims.n_type, E[2]), "emb"),
        (inv + "w1.weight", (Dl, Dl + 2), "w"),
        (inv + "w1.bias", (Dl,), "b:%d" % (Dl + 2)),
        (inv + "year_embedding.0.weight", (TIME_TABLE_ROWS[0], E[3]), "emb"),
        (inv + "month_embedding.0.weight", (TIME_TABLE_ROWS[1], E[3]), "emb"),
        (inv + "day_embedding.0.weight", (TIME_TABLE_ROWS[2], E[3]), "emb"),
        (inv + "hour_embedding.0.weight", (TIME_TABLE_ROWS[3], E[3]), "emb"),
    ]
    for name, D in (("label_attention", Dl), ("text_img_attention", P)):
        out += [
            (inv + name + ".mlp.fc1.weight", (D, 4 * D), "w"),
            (inv + name + ".mlp.fc1.bias", (D,), "b:%d" % (4 * D)),
            (inv + name + ".mlp.fc2.weight", (1, D), "w"),
            (inv + name + ".mlp.fc2.bias", (1,), "b:%d" % D),
        ]
    out += [
        ("instant_interest_model.out_fc.0.weight", (dims.instant_dim, 3), "w"),
        ("instant_interest_model.out_fc.0.bias", (dims.instant_dim,), "b:3"),
        ("bn.weight", (N,), "one"), ("bn.bias", (N,), "zero"),
        ("bn.running_mean", (N,), "zero"), ("bn.running_var", (N,), "one"),
        ("bn.num_batches_tracked", (), "count"),
    ]
    for name, od in (("gate", N), ("mlp", N), ("out_mlp", 1)):
        out += [
            (name + ".fc1.weight", (N // 4, N), "w"), (name + ".fc1.bias", (N // 4,), "b:%d" % N),
            (name + ".fc2.weight", (od, N // 4), "w"), (name + ".fc2.bias", (od,), "b:%d" % (N // 4)),
        ]
    if user_num is not None:
        out.append(("delta", (user_num + 1,), "delta"))
    return out


def make_state_dict(dims: Dims, seed: int = 1, user_num: int | None = None, perturb: bool = True):
    """Deterministic fp32 weights with PyTorch-default scales (U(+-1/sqrt(fan_in)) for Linear,
    N(0,1) for Embedding).  With ``perturb`` the BN affine/running stats and ``delta`` are moved
    off their trivial defaults so that parity tests exercise them."""
    rng = np.random.default_rng(seed)
    sd = {}
    for key, shape, kind in state_dict_shapes(dims, user_num):
        if kind == "emb":
            a = rng.standard_normal(shape)
        elif kind == "w":
            bound = 1.0 / np.sqrt(shape[-1])
            a = rng.uniform(-bound, bound, shape)
        elif kind.startswith("b:"):
            bound = 1.0 / np.sqrt(int(kind[2:]))
            a = rng.uniform(-bound, bound, shape)
        elif kind == "one":
            a = np.ones(shape) + (0.1 * rng.standard_normal(shape) if perturb else 0.0)
            if key.endswith("running_var"):
                a = np.abs(a) + 0.05
        elif kind == "zero":
            a = 0.1 * rng.standard_normal(shape) if perturb else np.zeros(shape)
        elif kind == "delta":
            a = 0.05 * rng.standard_normal(shape) if perturb else np.zeros(shape)
        elif kind == "count":
            sd[key] = np.array(0, dtype=np.int64)
            continue
        else:
            raise AssertionError(kind)
        sd[key] = np.ascontiguousarray(a, dtype=np.float32)
    return sd
