"""GPU: a time-boxed slice of the randomised stress (tests/stress_gpu.py) as a COLLECTED test, so that the driver's own run
-- not only the builder's log under profiles/ -- exercises random shapes: the three retrieval paths against each other
(bits), the sparse update against the oracle, every loss class dense and mined against the oracle, and the one-launch
step against the multi-kernel step (torch.equal).  ~20 s; the script itself runs for minutes."""
from __future__ import annotations

import pytest

pytestmark = pytest.mark.gpu


def test_randomised_stress_slice():
    from tests import stress_gpu

    n_ok, n_bad = stress_gpu.run(20.0, seed=20251005)
    assert n_bad == 0 and n_ok >= 40, (n_ok, n_bad)
