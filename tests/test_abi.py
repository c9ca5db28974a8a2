"""CPU: the C-ABI library loads, exports every symbol include/ddgani.h declares, and its host-side
entry points work without a GPU."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from ddgani_amd import _lib
from util_golden import load

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    txt = open(os.path.join(ROOT, "include", "ddgani.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(ddg_[a-z0-9_]+)\s*\(", txt)))


def test_header_symbols_exported_and_bound():
    syms = declared_symbols()
    assert len(syms) >= 15
    L = _lib.lib()
    for s in syms:
        assert hasattr(L, s), f"{s} declared in include/ddgani.h but not exported"
        assert s in _lib.SIGNATURES, f"{s} has no ctypes signature"
    assert sorted(_lib.SIGNATURES) == syms


def test_version_and_errors():
    L = _lib.lib()
    assert L.ddg_version() >= 100
    assert L.ddg_error_string(0) == b"ok"
    assert b"workspace" in L.ddg_error_string(-3)


@pytest.mark.parametrize("T", [6, 10])
def test_schedule_host(T):
    from ddgani_amd import ops
    z = load("schedule")
    s = ops.schedule_dict(T)
    for k, v in s.items():
        if k in ("sqrt_beta_s", "one_by_sqrt_alpha_s"):     # not used on the hot path; torch's vector sqrt is 1 ulp off
            assert np.allclose(v, z[f"T{T}_{k}"], rtol=1.2e-7, atol=0)
        else:
            assert np.array_equal(v, z[f"T{T}_{k}"]), (k, v, z[f"T{T}_{k}"])


def test_bad_args_rejected():
    L = _lib.lib()
    assert L.ddg_schedule(0, None) == -1
    assert L.ddg_attn_workspace_bytes(10, 10, 3, 1000, 5) == 0
    assert L.ddg_attn_workspace_bytes(1000, 1000, 8, 64, 5) > 0
