"""GPU: the SURVEY §8(f) rows -- test.py inference semantics, on-device per-row AUC / top-1, validation.
The model_test restatement in the oracle is 'parity unpinned' (test.py is not importable here, see its docstring);
the AUC kernel is checked against the oracle's AUC, which IS pinned to sklearn by make_golden.py."""
import numpy as np
import pytest
import torch

from golden_util import load_case, rel_err
from oracle import user_model_oracle as orc

pytestmark = pytest.mark.gpu


def test_row_auc_kernel_matches_oracle_with_ties_and_padding(lib):
    from news_recommendation_model_amd import evaluation
    rng = np.random.default_rng(0)
    B, T = 37, 21
    score = rng.standard_normal((B, T)).astype(np.float32)
    score[:, 3] = score[:, 5]                       # ties
    score[4] = 0.25                                 # all tied -> 0.5
    label = np.zeros((B, T), dtype=np.float32)
    label[np.arange(B), rng.integers(0, 9, B)] = 1
    live = rng.integers(9, T + 1, B)
    auc, top1 = evaluation.row_auc_top1(torch.from_numpy(score).cuda(), torch.from_numpy(label).cuda(),
                                        torch.from_numpy(live).cuda())
    auc, top1 = auc.cpu().numpy(), top1.cpu().numpy()
    for b in range(B):
        n = live[b]
        assert abs(auc[b] - orc.row_auc(label[b, :n], score[b, :n])) < 1e-6
        assert top1[b] == int(np.argmax(score[b, :n]) == np.argmax(label[b, :n]))
    # a row with one class only: sklearn raises; the kernel marks it with -1
    label[0] = 0
    auc, _ = evaluation.row_auc_top1(torch.from_numpy(score).cuda(), torch.from_numpy(label).cuda())
    assert float(auc[0]) == -1.0


def test_predict_follows_test_py_semantics(lib):
    """Two-model ensemble on a batch with per-row candidate padding: trimming, averaged softmax, second softmax."""
    from news_recommendation_model_amd import evaluation, synth, trainer
    case, dims, batch, sd, fx = load_case("tiny_pad")
    sd2 = synth.make_state_dict(dims, seed=5, user_num=int(batch["user_num"]))
    # rows get different amounts of padding: 2 (common, trimmed), 3, 2, 3
    batch["empty_num"][:] = [2, 3, 2, 3]
    batch["label"][:] = 0
    batch["label"][:, 0] = 1                          # the clicked candidate must survive every row's padding
    for b, z in enumerate(batch["empty_num"]):
        batch["x_target"][b, case["T"] - z:] = 0
        batch["x_global"][b, case["T"] - z:] = 0
    models = [trainer.build_model(dims, int(batch["user_num"]), s, device="cuda") for s in (sd, sd2)]
    tb = trainer.batch_to_device(batch, "cuda")
    scores, live = evaluation.predict(models, tb)
    tbc = {k: torch.from_numpy(v) for k, v in batch.items() if isinstance(v, np.ndarray) and v.ndim > 0}
    ref = orc.model_test_scores([orc.to_torch_params(s, False) for s in (sd, sd2)], tbc)
    assert scores.shape[1] == case["T"] - 2
    for b, r in enumerate(ref):
        assert int(live[b]) == len(r)
        assert rel_err(scores[b, :len(r)].cpu().numpy(), r) < 1e-3
    assert evaluation.rank_row([0.1, 0.7, 0.2]) == [3, 1, 2]
    # validation numbers (verify.py:19-43) against the oracle's AUC on the same scores
    auc_v, tpr_v = evaluation.validate(models, [tb])
    want_auc = np.mean([orc.row_auc(batch["label"][b, :len(r)], r) for b, r in enumerate(ref)])
    want_tpr = np.mean([float(np.argmax(r) == np.argmax(batch["label"][b])) for b, r in enumerate(ref)])
    assert abs(auc_v - want_auc) < 1e-5 and abs(tpr_v - want_tpr) < 1e-9


def test_checkpoint_roundtrip_drops_delta(lib, tmp_path):
    from news_recommendation_model_amd import evaluation, trainer
    case, dims, batch, sd, fx = load_case("tiny_train")
    model = trainer.build_model(dims, int(batch["user_num"]), sd, device="cuda")
    path = str(tmp_path / "ckpt_epoch_0.pth")
    evaluation.save_checkpoint(model, path)
    saved = torch.load(path, weights_only=True)
    assert "delta" not in saved and len(saved) == 37
    fresh = trainer.build_model(dims, 0, None, device="cuda")
    res = evaluation.load_checkpoint(fresh, path)
    assert res.missing_keys == ["delta"] and res.unexpected_keys == []
    assert torch.equal(fresh.gate.fc1.weight.cpu(), model.gate.fc1.weight.cpu())
