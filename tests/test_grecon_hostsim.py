"""The fused optimiser algorithm (glamr_amd/csrc/grecon_algo.hpp) run through the single-threaded host runtime of
tests/hostsim, checked against fixtures produced by the UNMODIFIED reference: first-iteration losses and gradients
(hand-written reverse pass vs PyTorch autograd) and the state after K Adam steps.  No GPU needed."""
import pytest

from oracle import make_golden as mg
from tests import grecon_common as gc


@pytest.mark.parametrize('cfg_id,T,P,K', mg.GRECON_CASES)
def test_fused_optimiser_vs_reference_fixture(asset_root, golden, cfg_id, T, P, K):
    gc.check_case(gc.hostsim_runner(), asset_root, golden, cfg_id, T, P, K)
