"""MI355X parity of the fused optimiser kernel (glamr_grecon_run_stage through the C ABI) against fixtures produced by the
UNMODIFIED reference: first-iteration losses and gradients, and the state after K Adam steps, for all six shipped configs."""
import pytest

from oracle import make_golden as mg
from tests import grecon_common as gc

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('cfg_id,T,P,K', mg.GRECON_CASES + mg.GRECON_CASES_WIDE)      # (_WIDE: 9 and 10 persons, csrc/grecon_wide.hip)
def test_fused_optimiser_kernel_vs_reference_fixture(asset_root, golden, cfg_id, T, P, K):
    gc.check_case(gc.device_runner(), asset_root, golden, cfg_id, T, P, K)


def test_poses_only_forward_pass_writes_the_same_world_poses(asset_root):
    """GLAMR_FLAG_POSES_ONLY through the C ABI (see tests/test_grecon_hostsim.py)."""
    gc.check_poses_only(gc.device_runner(), asset_root)
