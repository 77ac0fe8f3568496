"""Pin the oracle against the real reference and write tests/golden/*.npz.

Runs ONLY in the build container (it imports the reference's Python from /root/reference,
which never travels to the GPU box).  For every case it

  1. re-dimensions and constructs the reference ``UserModel`` (SURVEY.md §8c recipe), loads the
     deterministic weights of ``synth.make_state_dict``;
  2. runs the reference's step of train.py:66-75 (forward, loss, backward, torch.optim.Adam
     (lr 1e-3, weight_decay 1e-5) step) or an eval-mode forward;
  3. runs ``oracle/user_model_oracle.py`` on the same inputs and records the max abs difference
     (the oracle is "pinned" when these are at float rounding level);
  4. stores the reference's outputs as a fixture: inputs and weights are NOT stored, they are
     regenerated from seeds by ``news_recommendation_model_amd.synth`` (numpy PCG64, machine
     independent); a float64 checksum of both guards against drift.

Usage:  python oracle/make_golden.py            (rewrites tests/golden/)
"""
from __future__ import annotations

import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
REF = "/root/reference"

from news_recommendation_model_amd.config import Dims          # noqa: E402
from news_recommendation_model_amd import synth                 # noqa: E402
from oracle import user_model_oracle as orc                     # noqa: E402

# name -> case.  "full" stores every gradient / updated parameter; otherwise a strided sample.
CASES = {
    "tiny_train":  dict(B=4, H=8, T=5, emb=64, cat=50, mode="train", full=True),
    "tiny_eval":   dict(B=4, H=8, T=5, emb=64, cat=50, mode="eval", full=True),
    "tiny_pad":    dict(B=4, H=8, T=5, emb=64, cat=50, mode="train", full=True, pad_history=3, pad_target=2),
    "tiny_dupuser": dict(B=6, H=5, T=4, emb=64, cat=50, mode="train", full=True, dup_user=True),
    "odd_shape":   dict(B=3, H=19, T=7, emb=72, cat=40, mode="train", full=True),
    "refdefault":  dict(B=3, H=200, T=15, emb=64, cat=300, mode="train", full=False),
    "c1_demo":     dict(B=2, H=10, T=20, emb=256, cat=100, mode="train", full=False),
    "c2_small":    dict(B=2, H=32, T=30, emb=256, cat=100, mode="train", full=False),
    "c3_large":    dict(B=2, H=50, T=30, emb=400, cat=100, mode="train", full=False),
    "c5_long":     dict(B=1, H=128, T=64, emb=768, cat=100, mode="train", full=False),
}
# Eval-mode forward at BASELINE dimensions with padded candidates (round 5): model.eval() forward of the imported
# reference (models/user_model.py:27-35, running BatchNorm statistics) on candidate lists whose trailing rows are all-zero padding,
# as test.py:61's call sees them -- `r` and softmax(r, dim=1) (test.py:44,61: nn.Softmax(dim=1) of each model's logits) are stored.
# This pins the part of inference that CAN be pinned here; the trim / ensemble / second-softmax wrapper around it (test.py:48-56,
# :58-70) lives in a script that cannot be imported (zstandard) and stays "parity unpinned".
EVAL_CASES = {
    "refdefault_eval": dict(B=3, H=200, T=15, emb=64, cat=300, mode="eval", full=False, pad_history=40, pad_target=3),
    "c2_small_eval":   dict(B=3, H=32, T=30, emb=256, cat=100, mode="eval", full=False, pad_target=4),
    "c3_large_eval":   dict(B=3, H=50, T=30, emb=400, cat=100, mode="eval", full=False, pad_target=5),
}
# K-step trajectories of the reference's own training loop (train.py:66-75 repeated, torch.optim.Adam stepping every
# time): what pins BatchNorm running statistics, Adam moments and the -100 clamp regime of the loss beyond step 1.
# "fresh_batches": a new seeded batch per step (seed = step), else the same batch every step (as bench.py does).
TRAJ_CASES = {
    "traj_tiny": dict(B=4, H=8, T=5, emb=64, cat=50, steps=8, fresh_batches=True),
    "traj_c3":   dict(B=8, H=50, T=30, emb=400, cat=100, steps=8, fresh_batches=False),
}
SAMPLE = 1024
ZERO_GRAD_KEYS = ("delta", "out_mlp.fc2.bias")


def sample_idx(numel):
    if numel <= SAMPLE:
        return np.arange(numel)
    return np.unique(np.linspace(0, numel - 1, SAMPLE).astype(np.int64))


def case_inputs(case):
    dims = Dims.for_emb(case["emb"], category_label_num=case["cat"])
    batch = synth.make_batch(dims, case["B"], case["H"], case["T"], seed=0,
                             pad_history=case.get("pad_history", 0), pad_target=case.get("pad_target", 0))
    if case.get("dup_user"):
        batch["user_id"][:] = batch["user_id"][0]
        batch["user_id"][-1] = (batch["user_id"][0] + 1) % (int(batch["user_num"]) + 1)
    sd = synth.make_state_dict(dims, seed=1, user_num=int(batch["user_num"]))
    return dims, batch, sd


def checksum(arrs):
    return float(sum(np.asarray(a, dtype=np.float64).sum() for a in arrs))


def build_reference(dims):
    if REF not in sys.path:
        sys.path.insert(0, REF)
    import configs.model_config as mc
    from models.user_invariant_interest_model import UserInvariantInterestModel
    from models.user_model import UserModel
    mc.config["pca_vector"] = dims.pca_vector
    mc.config["category_label_num"] = dims.category_label_num
    UserInvariantInterestModel.__init__.__defaults__ = (list(dims.embed_setting),)
    return UserModel


def run_case(name, case, outdir):
    dims, batch, sd = case_inputs(case)
    UserModel = build_reference(dims)
    user_num = int(batch["user_num"])
    model = UserModel(user_num)
    missing = model.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in sd.items()}, strict=True)
    tb = {k: torch.from_numpy(v) for k, v in batch.items() if isinstance(v, np.ndarray)}
    fx = {}
    diffs = {}
    p = orc.to_torch_params(sd)

    if case["mode"] == "eval":
        model.eval()
        inter = {}
        hk = model.invariant_interest_model.register_forward_hook(lambda m, i, o: inter.__setitem__("inv", o))
        with torch.no_grad():
            r = model(tb["x_history"], tb["x_target"], tb["x_global"])
            loss = model.loss(tb["user_id"], r, tb["label"])
        hk.remove()
        with torch.no_grad():
            r_o, aux = orc.user_model_forward(p, tb["x_history"], tb["x_target"], tb["x_global"],
                                              training=False, return_aux=True)
            loss_o = orc.user_model_loss(p, tb["user_id"], r_o, tb["label"])
        diffs["r"] = float((r - r_o).abs().max())
        diffs["loss"] = float((loss - loss_o).abs())
        fx["r"] = r.numpy(); fx["loss"] = loss.numpy()
        fx["softmax_r"] = torch.nn.Softmax(dim=1)(r).numpy()                      # test.py:44,61
        diffs["eu_H"] = float((inter["inv"][0] - aux["eu_H"]).abs().max())
        fx["eu_H"] = inter["inv"][0].numpy(); fx["ec"] = inter["inv"][1].numpy()      # the REFERENCE's intermediates
    else:
        model.train()
        opt = torch.optim.Adam(model.parameters(), lr=1e-3, weight_decay=1e-5)   # train.py:48
        # hooks to grab the reference intermediates
        inter = {}
        h1 = model.invariant_interest_model.register_forward_hook(lambda m, i, o: inter.__setitem__("inv", o))
        h2 = model.invariant_interest_model.label_attention.register_forward_hook(lambda m, i, o: inter.__setitem__("s_lab", o))
        h3 = model.invariant_interest_model.text_img_attention.register_forward_hook(lambda m, i, o: inter.__setitem__("s_ti", o))
        r = model(tb["x_history"], tb["x_target"], tb["x_global"])
        loss = model.loss(tb["user_id"], r, tb["label"])
        loss.backward()
        grads = {k: (v.grad.detach().clone() if v.grad is not None else torch.zeros_like(v))
                 for k, v in model.named_parameters()}
        opt.step(); opt.zero_grad()
        for h in (h1, h2, h3):
            h.remove()
        after = {k: v.detach().clone() for k, v in model.state_dict().items()}

        opt_state = {"step": 0, "m": {}, "v": {}}
        tb_o = dict(tb)
        loss_o, r_o, grads_o = orc.train_step(p, opt_state, tb_o)
        diffs["r"] = float((r.detach() - r_o).abs().max())
        diffs["loss"] = float((loss.detach() - loss_o).abs())
        # delta and out_mlp.fc2.bias have a mathematically zero gradient (softmax over T is shift
        # invariant, models/user_model.py:38-41): only rounding noise remains, compared absolutely.
        diffs["grad_max_rel"] = max(
            float((grads[k] - grads_o[k]).abs().max() / (grads[k].abs().max() + 1e-30))
            for k in grads if k not in ZERO_GRAD_KEYS)
        diffs["zero_grad_max_abs"] = max(float(grads[k].abs().max()) for k in ZERO_GRAD_KEYS)
        diffs["param_after_max_abs"] = max(float((after[k].float() - p[k].detach().float()).abs().max()) for k in after)
        with torch.no_grad():
            _, aux = orc.user_model_forward(orc.to_torch_params(sd, False), tb["x_history"], tb["x_target"],
                                            tb["x_global"], training=True, return_aux=True)
        diffs["eu_H"] = float((inter["inv"][0].detach() - aux["eu_H"]).abs().max())
        diffs["score_label"] = float((inter["s_lab"].detach() - aux["score_label"]).abs().max())

        from tool.evaluation import auc_score                                       # tool/evaluation.py:3-5
        live_T = case["T"] - case.get("pad_target", 0)
        auc_ref = np.array([auc_score(batch["label"][b], r.detach().numpy()[b]) for b in range(case["B"])])
        auc_o = orc.batch_auc(batch["label"], r.detach().numpy())
        diffs["auc"] = float(np.abs(auc_ref - auc_o).max())

        fx["r"] = r.detach().numpy(); fx["loss"] = loss.detach().numpy(); fx["auc"] = auc_ref
        fx["eu_H"] = inter["inv"][0].detach().numpy(); fx["ec"] = inter["inv"][1].detach().numpy()
        if case["full"]:
            fx["score_label"] = inter["s_lab"].detach().numpy()
            fx["score_text_img"] = inter["s_ti"].detach().numpy()
        for k, g in grads.items():
            g = g.numpy().reshape(-1)
            idx = np.arange(g.size) if case["full"] else sample_idx(g.size)
            fx["grad/" + k] = g[idx]
            fx["gradnorm/" + k] = np.float64(np.linalg.norm(g.astype(np.float64)))
        for k, v in after.items():
            a = v.numpy().reshape(-1)
            idx = np.arange(a.size) if case["full"] else sample_idx(a.size)
            fx["after/" + k] = a[idx]

    fx["checksum_inputs"] = np.float64(checksum([batch["x_history"], batch["x_target"], batch["x_global"],
                                                 batch["label"], batch["user_id"]]))
    fx["checksum_weights"] = np.float64(checksum(sd.values()))
    np.savez_compressed(os.path.join(outdir, name + ".npz"), **fx)
    return diffs


def traj_batches(case, dims):
    user_num = 10 * case["B"]
    seeds = range(case["steps"]) if case["fresh_batches"] else [0] * case["steps"]
    return user_num, [synth.make_batch(dims, case["B"], case["H"], case["T"], seed=s, user_num=user_num) for s in seeds]


def run_trajectory(name, case, outdir):
    """K steps of the REFERENCE model + torch.optim.Adam(lr 1e-3, weight_decay 1e-5) (train.py:48,66-75); the oracle
    takes the same K steps beside it.  Stored per step: loss, logits (sample), max |logit|, BatchNorm running
    statistics; at the end: every parameter and both Adam moments (samples)."""
    dims = Dims.for_emb(case["emb"], category_label_num=case["cat"])
    user_num, batches = traj_batches(case, dims)
    sd = synth.make_state_dict(dims, seed=1, user_num=user_num, perturb=False)        # PyTorch-default BN / delta, as train.py:46
    UserModel = build_reference(dims)
    model = UserModel(user_num)
    model.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in sd.items()}, strict=True)
    model.train()
    opt = torch.optim.Adam(model.parameters(), lr=1e-3, weight_decay=1e-5)
    p = orc.to_torch_params(sd)
    ost = {"step": 0, "m": {}, "v": {}}
    K = case["steps"]
    fx = {"loss": np.zeros(K, np.float64), "maxlogit": np.zeros(K, np.float64)}
    rs, rms, rvs = [], [], []
    diffs = {"loss_rel": 0.0, "r_rel": 0.0, "bn_rel": 0.0}
    for step, batch in enumerate(batches):
        tb = {k: torch.from_numpy(v) for k, v in batch.items() if isinstance(v, np.ndarray) and v.ndim > 0}
        r = model(tb["x_history"], tb["x_target"], tb["x_global"])
        loss = model.loss(tb["user_id"], r, tb["label"])
        loss.backward()
        opt.step(); opt.zero_grad()
        loss_o, r_o, _ = orc.train_step(p, ost, tb)
        fx["loss"][step] = float(loss.detach())
        fx["maxlogit"][step] = float(r.detach().abs().max())
        rr = r.detach().numpy().reshape(-1)
        rs.append(rr[sample_idx(rr.size)])
        rms.append(model.bn.running_mean.numpy().copy()[sample_idx(model.bn.running_mean.numel())])
        rvs.append(model.bn.running_var.numpy().copy()[sample_idx(model.bn.running_var.numel())])
        diffs["loss_rel"] = max(diffs["loss_rel"], abs(float(loss.detach()) - float(loss_o)) / abs(float(loss.detach())))
        diffs["r_rel"] = max(diffs["r_rel"], float((r.detach() - r_o).abs().max() / r.detach().abs().max()))
        diffs["bn_rel"] = max(diffs["bn_rel"], float((model.bn.running_var - p["bn.running_var"]).abs().max()
                                                     / model.bn.running_var.abs().max()))
    fx["r"], fx["running_mean"], fx["running_var"] = np.stack(rs), np.stack(rms), np.stack(rvs)
    worst = 0.0
    for k, v in model.named_parameters():
        a = v.detach().numpy().reshape(-1)
        idx = sample_idx(a.size)
        fx["after/" + k] = a[idx]
        st = opt.state[v]
        fx["m/" + k] = st["exp_avg"].numpy().reshape(-1)[idx]
        fx["v/" + k] = st["exp_avg_sq"].numpy().reshape(-1)[idx]
        before = np.asarray(sd[k]).reshape(-1)[idx].astype(np.float64)
        move = np.linalg.norm(a[idx].astype(np.float64) - before)
        if move > 0 and k not in ZERO_GRAD_KEYS:       # those two follow the sign of rounding noise (gradient == 0 in exact arithmetic)
            worst = max(worst, float(np.linalg.norm(a[idx].astype(np.float64) - p[k].detach().numpy().reshape(-1)[idx]) / move))
    diffs["param_move_rel"] = worst          # |p_ref - p_oracle| / |p_ref - p_start|, worst tensor outside ZERO_GRAD_KEYS
    fx["checksum_inputs"] = np.float64(sum(checksum([b["x_history"], b["x_target"], b["x_global"], b["label"], b["user_id"]])
                                           for b in batches))
    fx["checksum_weights"] = np.float64(checksum(sd.values()))
    np.savez_compressed(os.path.join(outdir, name + ".npz"), **fx)
    return diffs


def attention_2d_case(outdir):
    """PointwiseAttentionExpanded with a 2-D target [B,D] (models/attention_model.py:64-65) and the
    stand-alone MLP, at D=64."""
    if REF not in sys.path:
        sys.path.insert(0, REF)
    from models.attention_model import PointwiseAttentionExpanded, MLP
    rng = np.random.default_rng(7)
    B, H, D = 3, 11, 64
    w = {"mlp.fc1.weight": rng.uniform(-.06, .06, (D, 4 * D)), "mlp.fc1.bias": rng.uniform(-.06, .06, (D,)),
         "mlp.fc2.weight": rng.uniform(-.12, .12, (1, D)), "mlp.fc2.bias": rng.uniform(-.12, .12, (1,))}
    w = {k: v.astype(np.float32) for k, v in w.items()}
    tgt = rng.standard_normal((B, D)).astype(np.float32)
    his = rng.standard_normal((B, H, D)).astype(np.float32)
    att = PointwiseAttentionExpanded(D)
    att.load_state_dict({k: torch.from_numpy(v) for k, v in w.items()})
    t_t = torch.from_numpy(tgt).requires_grad_(True)
    h_t = torch.from_numpy(his).requires_grad_(True)
    s = att(t_t, h_t)
    gs = torch.from_numpy(rng.standard_normal(tuple(s.shape)).astype(np.float32))
    (s * gs).sum().backward()
    p = {"a." + k: torch.from_numpy(v) for k, v in w.items()}
    s_o = orc.pointwise_attention_scores(p, "a", torch.from_numpy(tgt), torch.from_numpy(his))
    fx = {"target": tgt, "history": his, "grad_scores": gs.numpy(), "scores": s.detach().numpy(),
          "grad_target": t_t.grad.numpy(), "grad_history": h_t.grad.numpy()}
    for k, v in w.items():
        fx["w/" + k] = v
    for k, v in att.named_parameters():
        fx["grad/" + k] = v.grad.numpy()
    # stand-alone MLP(264 -> 264) on random rows
    m = MLP(264, 264)
    mw = {k: rng.uniform(-.06, .06, tuple(v.shape)).astype(np.float32) for k, v in m.state_dict().items()}
    m.load_state_dict({k: torch.from_numpy(v) for k, v in mw.items()})
    x = rng.standard_normal((17, 264)).astype(np.float32)
    fx["mlp_x"] = x
    fx["mlp_y"] = m(torch.from_numpy(x)).detach().numpy()
    for k, v in mw.items():
        fx["mlp_w/" + k] = v
    np.savez_compressed(os.path.join(outdir, "attention_2d.npz"), **fx)
    return {"scores": float((s.detach() - s_o).abs().max())}


def degenerate_case(outdir):
    """What the REFERENCE does with no rows: empty history (a real forward), empty batch (eval: empty result,
    train: BatchNorm raises), a single training row (BatchNorm raises), BCELoss over nothing (nan)."""
    dims = Dims.for_emb(16, 40)
    UserModel = build_reference(dims)
    sd = synth.make_state_dict(dims, seed=3, user_num=5)
    model = UserModel(5)
    model.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, strict=False)
    batch = synth.make_batch(dims, 3, 2, 4, seed=11, user_num=5)
    xh = torch.from_numpy(batch["x_history"]).float()
    xt = torch.from_numpy(batch["x_target"]).float()
    xg = torch.from_numpy(batch["x_global"]).float()
    beh = {}

    def outcome(fn):
        try:
            out = fn()
            return "shape:" + "x".join(str(int(d)) for d in out.shape)
        except Exception as e:                                     # noqa: BLE001 -- the exception type IS the datum
            return "raises:" + type(e).__name__
    for mode in ("train", "eval"):
        model.train(mode == "train")
        with torch.no_grad():
            beh["empty_history_" + mode] = outcome(lambda: model(xh[:, :0], xt, xg))
            beh["empty_batch_" + mode] = outcome(lambda: model(xh[:0], xt[:0], xg[:0]))
            beh["no_candidates_" + mode] = outcome(lambda: model(xh, xt[:, :0], xg[:, :0]))
            beh["single_row_" + mode] = outcome(lambda: model(xh[:1], xt[:1, :1], xg[:1, :1]))
    att = model.invariant_interest_model.text_img_attention
    D = dims.pca_vector
    for name, (B, T, H) in {"B0": (0, 3, 4), "T0": (2, 0, 4), "H0": (2, 3, 0)}.items():
        beh["attention_" + name] = outcome(lambda: att(torch.zeros(B, T, D), torch.zeros(B, H, D)))
    beh["empty_batch_loss"] = outcome(lambda: model.loss(torch.zeros(0, dtype=torch.long), torch.zeros(0, 4), torch.zeros(0, 4)))
    loss = model.loss(torch.zeros(0, dtype=torch.long), torch.zeros(0, 4), torch.zeros(0, 4))
    beh["empty_batch_loss_is_nan"] = bool(torch.isnan(loss))
    return beh


def main():
    if "--only-traj" in sys.argv:                                   # add / refresh the trajectory fixtures only
        torch.manual_seed(0)
        torch.set_num_threads(8)
        outdir = os.path.join(ROOT, "tests", "golden")
        with open(os.path.join(outdir, "MANIFEST.json")) as f:
            manifest = json.load(f)
        manifest["trajectories"] = {}
        for name, case in TRAJ_CASES.items():
            diffs = run_trajectory(name, case, outdir)
            manifest["trajectories"][name] = {"case": case, "oracle_vs_reference": diffs}
            print(name, diffs, flush=True)
        with open(os.path.join(outdir, "MANIFEST.json"), "w") as f:
            json.dump(manifest, f, indent=1, sort_keys=True)
        return
    if "--only-eval" in sys.argv:                                   # add / refresh the BASELINE-dims eval fixtures only
        torch.manual_seed(0)
        torch.set_num_threads(8)
        outdir = os.path.join(ROOT, "tests", "golden")
        with open(os.path.join(outdir, "MANIFEST.json")) as f:
            manifest = json.load(f)
        for name, case in EVAL_CASES.items():
            diffs = run_case(name, case, outdir)
            manifest["cases"][name] = {"case": case, "oracle_vs_reference": diffs}
            print(name, diffs, flush=True)
        with open(os.path.join(outdir, "MANIFEST.json"), "w") as f:
            json.dump(manifest, f, indent=1, sort_keys=True)
        return
    if "--only-degenerate" in sys.argv:                             # add / refresh that one entry, leave the rest alone
        outdir = os.path.join(ROOT, "tests", "golden")
        with open(os.path.join(outdir, "MANIFEST.json")) as f:
            manifest = json.load(f)
        manifest["degenerate"] = degenerate_case(outdir)
        print("degenerate", manifest["degenerate"])
        with open(os.path.join(outdir, "MANIFEST.json"), "w") as f:
            json.dump(manifest, f, indent=1, sort_keys=True)
        return
    torch.manual_seed(0)
    torch.set_num_threads(8)
    outdir = os.path.join(ROOT, "tests", "golden")
    os.makedirs(outdir, exist_ok=True)
    manifest = {"generator": "oracle/make_golden.py", "reference": "ChuhanZhou/News_Recommendation_Model @ 2024-12-18",
                "torch": torch.__version__, "numpy": np.__version__, "cases": {}}
    for name, case in {**CASES, **EVAL_CASES}.items():
        diffs = run_case(name, case, outdir)
        manifest["cases"][name] = {"case": case, "oracle_vs_reference": diffs}
        print(name, diffs, flush=True)
    manifest["cases"]["attention_2d"] = {"case": {"B": 3, "H": 11, "D": 64}, "oracle_vs_reference": attention_2d_case(outdir)}
    print("attention_2d", manifest["cases"]["attention_2d"]["oracle_vs_reference"])
    manifest["degenerate"] = degenerate_case(outdir)
    print("degenerate", manifest["degenerate"])
    manifest["trajectories"] = {}
    for name, case in TRAJ_CASES.items():
        diffs = run_trajectory(name, case, outdir)
        manifest["trajectories"][name] = {"case": case, "oracle_vs_reference": diffs}
        print(name, diffs, flush=True)
    with open(os.path.join(outdir, "MANIFEST.json"), "w") as f:
        json.dump(manifest, f, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
