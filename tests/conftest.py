import os
import sys
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')
    config.addinivalue_line('markers', 'reference: needs /root/reference (build container only)')


@pytest.fixture(scope='session')
def asset_root(tmp_path_factory):
    """Synthetic SMPL model + checkpoints regenerated from seeds (identical on every machine)."""
    from oracle.port import build
    root = os.environ.get('GLAMR_ASSET_ROOT') or str(tmp_path_factory.mktemp('assets'))
    return build.ensure_synthetic_assets(root)


@pytest.fixture(scope='session')
def golden():
    import numpy as np
    gdir = os.path.join(ROOT, 'tests', 'golden')

    def load(name):
        return dict(np.load(os.path.join(gdir, name + '.npz'), allow_pickle=False))
    return load
