"""CPU: the dependency search of the "data-dependency" plug-in (ddgani_amd/model/FD_model.py: discover_FDs, the
vectorised restatement of get_FD_model_Tree / DFS / isCurNodeFD, model/FD_model.py:113-164, :232-307) against the
functional dependencies the REFERENCE finds on its `hospital` table (tests/golden/fd_hospital.npz, produced by
tests/golden/make_fd_golden.py by running the reference)."""
import numpy as np
import pandas as pd

from util_golden import load


def test_discovered_dependencies_equal_the_references():
    from ddgani_amd.model import FD_model as F
    from ddgani_amd.utils.util import sort_corr
    z = load("fd_hospital")
    fds = F.discover_FDs(pd.DataFrame(z["filled_coded"]), z["data_m"], z["categorical_cols"].tolist(), sort_corr(z["corr"]))
    ref = [([int(v) for v in xs if v >= 0], int(y)) for xs, y in zip(z["fd_x"], z["fd_y"])]
    assert fds == ref and len(ref) == int(z["n_fd"])


def test_fd_check_and_bookkeeping():
    from ddgani_amd.model import FD_model as F
    # y = f(a), y not a function of b; (a, b) trivially determines y but {a} is already known
    a = np.array([0, 0, 1, 1, 2, 2, 0, 1])
    b = np.array([0, 1, 0, 1, 0, 1, 1, 0])
    y = np.array([5, 5, 6, 6, 5, 5, 5, 6])
    obs = [np.ones(8, dtype=bool)] * 3
    enc = [a, b, y]
    assert F.isCurNodeFD(enc, 2, obs, [0]) and not F.isCurNodeFD(enc, 2, obs, [1]) and F.isCurNodeFD(enc, 2, obs, [1, 0])
    assert not F.isCurNodeFD(enc, 2, obs, [])
    none = [np.zeros(8, dtype=bool)] * 3
    assert not F.isCurNodeFD(enc, 2, none, [0])                    # no co-observed rows
    feat = [[1, 0]]
    F.updateFeatLabels([0], feat)                                  # a subset replaces its supersets
    assert feat == [[0]] and F.isCurNodeInFDs([3, 0], feat) and not F.isCurNodeInFDs([3], feat)
    assert abs(F.entropy(np.array([1, 1, 2, 2])) - 1.0) < 1e-12 and F.entropy(np.array([])) == 0


def test_numeric_column_codes_equal_the_references_loop():
    """_encode_columns' numeric branch (two binary searches per value) == the reference's ten np.isin assignments
    (model/FD_model.py:140-147 of the reference: O(N^2 / 10), 13 minutes at 200 000 rows) on columns with duplicates"""
    from ddgani_amd.model import FD_model as F
    rng = np.random.default_rng(1)
    for n in (7, 10, 23, 1000, 3001):
        arr = np.round(rng.normal(size=n), 1)
        df = pd.DataFrame({0: arr, 1: rng.integers(0, 3, n).astype(str)})
        enc = F._encode_columns(df, [1])[0]
        ref = np.zeros_like(arr)
        for i, b in enumerate(np.array_split(np.sort(arr), 10)):
            ref[np.isin(arr, b)] = i
        assert np.array_equal(np.asarray(enc, dtype=float), ref), n
