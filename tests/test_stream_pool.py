"""A closed context's streams are taken over by the next context of the process (capi.hip: stream_set_acquire): the stream ->
hardware queue mapping a model runs on does not depend on how many models the process has opened and closed before it.
Contexts that are alive together never share streams."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def test_second_context_takes_over_the_streams_of_a_closed_one():
    from gpcsd_amd import _hip
    a = _hip.Context()
    ha = a.stream_handles()
    assert len(set(ha)) == 4 and all(ha)
    b = _hip.Context()                                     # alive together: a set of its own
    hb = b.stream_handles()
    assert not set(ha) & set(hb)
    created0, reused0 = a.stream_pool_stats()
    a.close()
    c = _hip.Context()                                     # takes over a's streams, role by role
    assert c.stream_handles() == ha
    created1, reused1 = c.stream_pool_stats()
    assert created1 == created0 and reused1 == reused0 + 1
    # the taken-over streams work: an eigen-decomposition on c agrees with one on b, bit for bit
    rng = np.random.default_rng(5)
    m = rng.standard_normal((300, 300))
    m = m @ m.T
    wc, vc = c.eigh(m)
    wb, vb = b.eigh(m)
    assert np.array_equal(wc, wb) and np.array_equal(vc, vb)
    assert np.abs(np.sort(wc) - np.linalg.eigvalsh(m)).max() <= 1e-12 * np.abs(wc).max()
    b.close()
    c.close()


def test_models_opened_one_after_another_run_on_the_same_streams():
    from gpcsd_amd import _hip
    from gpcsd_amd.gpcsd1d import GPCSD1D
    handles = []
    for seed in range(3):
        rng = np.random.default_rng(seed)
        x = np.linspace(0.0, 2300.0, 24)[:, None]
        t = np.arange(60.0)[:, None]
        m = GPCSD1D(rng.standard_normal((24, 60, 4)), x, t)
        ll = m.loglik()
        assert np.isfinite(ll)
        handles.append(m._context().stream_handles())
        m._context().close()                               # (what dropping the last reference to the model does)
        del m
    assert handles[0] == handles[1] == handles[2]
