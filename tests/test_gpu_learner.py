"""ddg_learner_train (through the C-ABI) against fixtures produced by the reference's
model/Learner.py::train_L_code (tests/golden/make_learner_golden.py) and against the oracle restatement."""
import numpy as np
import pytest
import torch

from oracle import ddgani_oracle as orc
from util_golden import load

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    return torch.device("cuda:0")


@pytest.mark.parametrize("name", ["learner_small", "learner_wide", "learner_manyclass"])
def test_learner_golden(dev, name):
    from ddgani_amd import ops
    z = load(name)
    L, epoch = int(z["L"]), int(z["epoch"])
    W0, b0 = orc.learner_init(z["x_train"].shape[1], L)
    W, b = W0.to(dev).contiguous(), b0.to(dev).contiguous()
    T = lambda a: torch.tensor(a).to(dev)
    loss, acc = ops.learner_train(T(z["x_train"]), T(z["y_train"]), T(z["x_valid"]), T(z["y_valid"]), W, b, epoch)
    # fp32 summation order differs from the reference's GEMMs; SGD on this smooth problem does not amplify it
    assert abs(float(loss) - float(z["val_loss"])) < 2e-5
    assert abs(acc - float(z["acc"])) <= 1.0 / z["x_valid"].shape[0] + 1e-12
    np.testing.assert_allclose(W.cpu().numpy(), z["W"], rtol=0, atol=2e-5)
    np.testing.assert_allclose(b.cpu().numpy(), z["b"], rtol=0, atol=2e-5)


def test_learner_mirror_reseeds_rng(dev):
    """train_L_code keeps the reference's side effect: torch's global RNG is re-seeded to 3407 and advanced by the
    construction of L (model/Learner.py:44-45)."""
    from ddgani_amd.model.Learner import train_L_code
    z = load("learner_small")
    T = lambda a: torch.tensor(a).to(dev)
    torch.manual_seed(1)
    loss, acc = train_L_code(50, T(z["x_train"]), T(z["y_train"])[:, None], T(z["x_valid"]), T(z["y_valid"])[:, None],
                             int(z["L"]), dev)
    after = torch.rand(3)
    torch.manual_seed(3407)
    D = z["x_train"].shape[1]
    torch.nn.Linear(D, D); torch.nn.Linear(D, int(z["L"])); torch.nn.Linear(D // 3, int(z["L"]))
    torch.nn.BatchNorm1d(D)
    assert torch.equal(after, torch.rand(3))
    o_loss, o_acc, _, _ = orc.learner_train(50, torch.tensor(z["x_train"]), torch.tensor(z["y_train"]),
                                            torch.tensor(z["x_valid"]), torch.tensor(z["y_valid"]), int(z["L"]))
    assert abs(float(loss) - o_loss) < 1e-5 and abs(acc - o_acc) < 1e-9


def test_learner_large_rows(dev):
    """200k x 176 rows (the c3 width): the fused step against the oracle's explicit arithmetic on the host."""
    from ddgani_amd import ops
    rng = np.random.default_rng(3)
    n, D, L = 200_000, 176, 5
    centers = rng.normal(size=(L, D)).astype(np.float32) * 0.3
    y = rng.integers(0, L, n)
    x = (centers[y] + rng.normal(size=(n, D))).astype(np.float32)
    W0, b0 = orc.learner_init(D, L)
    W, b = W0.to(dev).contiguous(), b0.to(dev).contiguous()
    T = lambda a: torch.tensor(a).to(dev)
    loss, acc = ops.learner_train(T(x), T(y), T(x[:5000]), T(y[:5000]), W, b, 20)
    o_loss, o_acc, oW, ob = orc.learner_train(20, torch.tensor(x), torch.tensor(y), torch.tensor(x[:5000]),
                                              torch.tensor(y[:5000]), L)
    assert abs(float(loss) - o_loss) < 1e-5
    np.testing.assert_allclose(W.cpu().numpy(), oW.numpy(), rtol=0, atol=1e-5)
