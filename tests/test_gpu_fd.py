"""GPU: the classifiers of the "data-dependency" plug-in through the C-ABI (ddg_fd_train_epoch / ddg_fd_eval) against
the REFERENCE's own training of its first three dependencies on `hospital` (tests/golden/fd_hospital.npz: the rows
X / Y it built, its initial weights, the state of torch's CPU generator before train_Model, its trained weights, and
get_my_FD_loss of the three models on a batch).  The shuffles replay from the generator state, so the run walks
the reference's batches; what differs is fp32 summation order over up to 1 200 SGD steps: trained weights within
2e-3 relative, the plug-in's loss within 1e-3 relative."""
import numpy as np
import pytest
import torch

from util_golden import load

pytestmark = pytest.mark.gpu


class _F:
    def __init__(self, w):
        self.data_type = "Categorical Data" if w > 1 else "Numerical Data"
        self.dict = {i: i for i in range(w)}

    def dim(self):
        return len(self.dict)


def test_fd_classifier_training_and_loss_match_reference():
    from ddgani_amd.model import FD_model as F
    dev = torch.device("cuda:0")
    z = load("fd_hospital")
    fields = [_F(int(w)) for w in z["widths"]]
    models = []
    for i in range(int(z["n_trained"])):
        X, Y = torch.tensor(z[f"t{i}__X"]).to(dev), torch.tensor(z[f"t{i}__Y"]).to(dev)
        xs = [int(v) for v in z["fd_x"][i] if v >= 0]
        m = F.FDModel(X.shape[1], int(z[f"t{i}__W0__fc3__weight"].shape[0]), xs, int(z["fd_y"][i]))
        m.load_state_dict({k: torch.tensor(z[f"t{i}__W0__" + k.replace(".", "__")]) for k in m.state_dict()})
        m.to(dev)
        torch.set_rng_state(torch.tensor(z[f"t{i}__rng_state"]))
        F.train_Model(X, Y, m)
        for k in ("fc1.weight", "fc1.bias", "fc3.weight", "fc3.bias"):
            ref = z[f"t{i}__W1__" + k.replace(".", "__")]
            got = m.state_dict()[k].cpu().numpy()
            rel = np.abs(got - ref).max() / np.abs(ref).max()
            assert rel < 2e-3, (i, k, rel)
        assert np.array_equal(m.state_dict()["fc2.weight"].cpu().numpy(), z[f"t{i}__W0__fc2__weight"])   # never trained
        # the same rows through the reference-trained weights: accuracy and loss of both agree
        models.append(m)
    batch = torch.tensor(z["x"][:128]).to(dev)
    loss = float(F.get_my_FD_loss(models, batch, fields))
    ref = float(z["fd_loss_batch128"])
    assert abs(loss - ref) < 1e-3 * abs(ref), (loss, ref)
    # the loss on REFERENCE-trained weights (forward + cross entropy only): 1e-5
    for i, m in enumerate(models):
        sd = {k: torch.tensor(z[f"t{i}__W1__" + k.replace(".", "__")]) for k in m.state_dict()}
        m.load_state_dict(sd)
        m.to(dev)
        m.device_params(refresh=True)
    loss = float(F.get_my_FD_loss(models, batch, fields))
    assert abs(loss - ref) < 1e-5 * abs(ref), (loss, ref)


def test_fd_eval_against_torch():
    """ddg_fd_eval on random data vs the same arithmetic in torch (fp64), incl. chunked evaluation and ties"""
    from ddgani_amd import ops
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(1)
    n, din, dout = 3000, 37, 5
    X = torch.randn(n, din, generator=g)
    Y = torch.randint(0, dout, (n,), generator=g)
    W1, b1 = torch.randn(din, din, generator=g) * 0.2, torch.randn(din, generator=g) * 0.1
    W3, b3 = torch.randn(dout, din, generator=g) * 0.2, torch.randn(dout, generator=g) * 0.1
    P = ops.FdParams(W1.to(dev), b1.to(dev), W3.to(dev), b3.to(dev))
    o = torch.relu(X.double() @ W1.double().T + b1.double()) @ W3.double().T + b3.double()
    ce = torch.nn.functional.cross_entropy(o, Y).item()
    hits = int((o.argmax(1) == Y).sum())
    for chunk in (65536, 1000, 777):
        h, c = ops.fd_eval(P, X.to(dev), Y.to(torch.int32).to(dev), chunk_rows=chunk)
        assert int(h.item()) == hits and abs(float(c.item()) - ce) < 2e-6 * max(1, abs(ce))
    # one SGD epoch == torch autograd on the same batches
    perm = torch.randperm(n, generator=g)
    W = [t.clone().requires_grad_(True) for t in (W1, b1, W3, b3)]
    for r0 in range(0, n, 256):
        idx = perm[r0:r0 + 256]
        out = torch.relu(X[idx] @ W[0].T + W[1]) @ W[2].T + W[3]
        loss = torch.nn.functional.cross_entropy(out, Y[idx])
        gr = torch.autograd.grad(loss, W)
        with torch.no_grad():
            for p_, g_ in zip(W, gr):
                p_ -= 0.02 * g_
    ops.fd_train_epoch(P, X.to(dev), Y.to(torch.int32).to(dev), perm.to(dev), batch=256, lr=0.02)
    for got, ref in zip((P.W1, P.b1, P.W3, P.b3), W):
        assert (got.cpu() - ref.detach()).abs().max().item() < 2e-6


def test_fd_plugin_at_200k_rows_mixed_table():
    """--UseFD beyond the 1 000-row `hospital`: a 200 000-row mixed table (4 categorical + 2 numeric columns, MCAR 0.2)
    with planted dependencies  A -> B (B = A mod 4),  A -> D (D = A div 4),  (B, D) -> A,  and an independent column C.
    get_FD_model_Tree = the host discovery (np.unique groupings, model/FD_model.py:279-307 of the reference) + one
    classifier per dependency trained on the GPU (ddg_fd_train_epoch): the discovered set is exactly the planted one,
    every classifier reaches accuracy 1 on its observed rows (the early exit of train_Model), and get_my_FD_loss on a
    batch equals a plain torch forward of the trained weights to 1e-5.
    (BASELINE config 5's synthetic table is all-numeric, so --UseFD discovers nothing there: the plug-in only takes
    categorical right-hand sides, model/FD_model.py:120-128 of the reference.)"""
    import pandas as pd
    from ddgani_amd.model import FD_model as F
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(4)
    N = 200_000
    A = rng.integers(0, 12, N)
    cols = [A, A % 4, rng.integers(0, 6, N), A // 4, rng.normal(size=N), rng.uniform(size=N)]
    cards = [12, 4, 6, 3, 1, 1]
    cat = [0, 1, 2, 3]
    data_m = (rng.uniform(size=(N, 6)) < 0.8).astype(np.int64)
    miss = pd.DataFrame({i: (np.where(data_m[:, i] == 1, cols[i].astype(str), "Null") if i in cat
                             else np.where(data_m[:, i] == 1, cols[i], 0.0)) for i in range(6)})
    # association order (what sort_corr would give): any order works for the planted set; use descending index distance
    corr = np.abs(rng.uniform(size=(6, 6)))
    sort_corr = {i: sorted(range(6), key=lambda j: corr[i][j], reverse=True) for i in range(6)}
    fields = [_F(c) for c in cards]
    code = np.zeros((N, sum(cards)), dtype=np.float32)
    off = np.concatenate([[0], np.cumsum(cards)])
    for i in range(6):
        if i in cat:
            code[np.arange(N), off[i] + cols[i]] = 1.0
        else:
            code[:, off[i]] = cols[i]
    M = np.repeat(data_m, cards, axis=1).astype(np.float32)
    zero_feed = torch.tensor(code * M).to(dev)
    torch.manual_seed(0)
    models = F.get_FD_model_Tree(miss, data_m, cat, zero_feed, fields, dev, sort_corr)
    found = sorted((sorted(m.x_index_list), m.y_index) for m in models)
    assert found == [([0], 1), ([0], 3), ([1, 3], 0)], found
    full = torch.tensor(code).to(dev)
    for m in models:
        rows = np.ones(N, dtype=bool)
        for c in m.x_index_list + [m.y_index]:
            rows &= data_m[:, c] == 1
        X, Y = F._fd_input(full[torch.as_tensor(np.nonzero(rows)[0], device=dev)], fields, m.x_index_list, m.y_index)
        assert F.test_model(X, Y.to(torch.int32), m) == 1.0, (m.x_index_list, m.y_index)
    batch = full[:4096]
    loss = float(F.get_my_FD_loss(models, batch, fields))
    ref = 0.0
    for m in models:
        X, Y = F._fd_input(batch, fields, m.x_index_list, m.y_index)
        sd = {k: v.to(dev).double() for k, v in m.state_dict().items()}
        h = torch.relu(X.double() @ sd["fc1.weight"].T + sd["fc1.bias"])
        logits = h @ sd["fc3.weight"].T + sd["fc3.bias"]
        ref += float(torch.nn.functional.cross_entropy(logits, Y))
    assert abs(loss - ref) <= 1e-5 * max(1.0, abs(ref)), (loss, ref)
