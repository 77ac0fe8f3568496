"""CPU: the Module mirror keeps the reference's interface (names, state_dict keys and shapes,
attributes), is picklable, loads the reference's shipped checkpoints, and refuses to compute on CPU."""
import os
import pickle

import numpy as np
import pytest
import torch

import news_recommendation_model_amd as nrm
from news_recommendation_model_amd import synth, trainer
from news_recommendation_model_amd.config import Dims


def test_state_dict_keys_and_shapes_match_reference_layout():
    dims = Dims()                                          # reference defaults
    model = nrm.UserModel(7)
    want = {k: tuple(s) for k, s, _ in synth.state_dict_shapes(dims, user_num=7)}
    got = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    assert got == want
    assert len(got) == 38                                  # 37 + delta
    assert model.bn.num_features == 264 and model.delta.shape == (8,)
    assert model.invariant_interest_model.embed_setting == [32, 16, 8, 8]
    assert model.instant_interest_model.output_dim == 8


@pytest.mark.parametrize("emb", [72, 256, 400])
def test_redimensioned_model(emb):
    dims = Dims.for_emb(emb, category_label_num=30)
    model = trainer.build_model(dims, 5, device="cpu")
    want = {k: tuple(s) for k, s, _ in synth.state_dict_shapes(dims, user_num=5)}
    assert {k: tuple(v.shape) for k, v in model.state_dict().items()} == want
    sd = synth.make_state_dict(dims, seed=3, user_num=5)
    model.load_state_dict({k: torch.as_tensor(v) for k, v in sd.items()})
    np.testing.assert_array_equal(model.gate.fc1.weight.detach().numpy(), sd["gate.fc1.weight"])
    # the global config is restored after construction
    assert nrm.model_config["pca_vector"] == 64


def test_forward_on_cpu_is_refused_not_emulated():
    model = nrm.UserModel(3)
    xh = torch.zeros(2, 5, 80, dtype=torch.float64)
    xt = torch.zeros(2, 3, 78, dtype=torch.float64)
    with pytest.raises(RuntimeError, match="MI355X"):
        model(xh, xt, torch.zeros(2, 3, 3))
    with pytest.raises(RuntimeError, match="MI355X"):
        nrm.PointwiseAttentionExpanded(64)(torch.zeros(2, 64), torch.zeros(2, 4, 64))
    with pytest.raises(RuntimeError, match="MI355X"):
        nrm.MLP(64, 8)(torch.zeros(3, 64))
    with pytest.raises(RuntimeError, match="MI355X"):
        model.loss(torch.tensor([0, 1]), torch.zeros(2, 3), torch.zeros(2, 3))


def test_pickle_roundtrip_like_test_py_child_process():
    # reference test.py:177 pickles the CPU model list into a child process
    model = nrm.UserModel(3)
    clone = pickle.loads(pickle.dumps([model]))[0]
    for (k, a), (_, b) in zip(model.state_dict().items(), clone.state_dict().items()):
        assert torch.equal(a, b), k


@pytest.mark.parametrize("name", ["ckpt_ebnerd_large_train_final.pth", "ckpt_ebnerd_large_validation_final.pth"])
def test_reference_checkpoints_load(name):
    path = os.path.join("/root/reference/ckpt", name)
    if not os.path.exists(path):
        pytest.skip("reference checkpoints are only present in the build container")
    sd = torch.load(path, weights_only=True, map_location="cpu")
    model = nrm.UserModel()
    res = model.load_state_dict(sd, strict=False)          # reference test.py:160
    assert res.unexpected_keys == [] and res.missing_keys == ["delta"]
