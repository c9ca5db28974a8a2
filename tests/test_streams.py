"""Random-stream ids of the on-chip generators (ddgani_amd/model/Diffusion.py: stream_of): no two purposes,
timesteps, ranks or calls share a Philox stream, also after the library's own derivations (ddg_denoiser_fwd uses
stream*4 + layer, the training step stream*8 + 7 for eps and stream*4 + layer for dropout)."""
import numpy as np
import pytest


def test_stream_ids_do_not_collide():
    from ddgani_amd.model import Diffusion as D_
    ids = []
    for call in (1, 2, 3, 1000, (1 << 36) - 1):
        for kind in (D_.KIND_EPS, D_.KIND_Z, D_.KIND_EPS2, D_.KIND_DROP):
            for rank in (0, 1, 7, 255):
                ts = np.unique(np.concatenate([np.arange(0, 300), np.array([1000, D_.MAX_T])]))
                ids.append(np.array([D_.stream_of(call, kind, int(t), rank) for t in ts], dtype=np.uint64))
    ids = np.concatenate(ids)
    assert ids.size == np.unique(ids).size
    # library-side derivations stay distinct and inside 64 bits
    drop = np.concatenate([ids * np.uint64(4) + np.uint64(l) for l in range(3)])
    eps = ids * np.uint64(8) + np.uint64(7)
    assert int(ids.max()) * 8 + 7 < 1 << 64
    both = np.concatenate([drop, eps])
    assert both.size == np.unique(both).size
    with pytest.raises(ValueError):
        D_.stream_of(1, D_.KIND_Z, D_.MAX_T + 1)
    with pytest.raises(ValueError):
        D_.stream_of(1, D_.KIND_Z, 3, rank=256)
    assert D_.drop_seed(0) != 0 and D_.drop_seed(D_.drop_seed(12345)) == 12345
