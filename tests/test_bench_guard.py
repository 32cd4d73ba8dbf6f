"""bench.py's wall-clock guard around the one multi-rank line with data-path collectives (person-sharded configs[3]): a result, an exception and a
call that never returns must each leave the headline intact."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def test_guarded_returns_the_result():
    assert bench.guarded(lambda: {'x': 1}, torch.device('cpu'), 5.0) == ({'x': 1}, None)


def test_guarded_reports_an_exception_instead_of_raising():
    def boom():
        raise RuntimeError('peer lost')
    res, why = bench.guarded(boom, torch.device('cpu'), 5.0)
    assert res is None and 'RuntimeError' in why and 'peer lost' in why


def test_guarded_gives_up_on_a_call_that_never_returns():
    t0 = time.time()
    res, why = bench.guarded(lambda: time.sleep(60), torch.device('cpu'), 0.3)
    assert res is None and 'no result after' in why and time.time() - t0 < 5.0
