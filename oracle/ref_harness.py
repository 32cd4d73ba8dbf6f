"""TEST INFRASTRUCTURE ONLY -- runs the UNMODIFIED reference (/root/reference) in the build container.

The reference cannot be imported as shipped: `smplx` and `pytorch_lightning` are absent.  This harness
  1. puts two import stubs on sys.path (oracle/refstubs: `pytorch_lightning` = nn.Module + load_from_checkpoint;
     `smplx` = oracle/smplx_lbs.py, a restatement of the published SMPL arithmetic),
  2. builds a scratch working directory whose `global_recon lib motion_infiller traj_pred` entries are symlinks into
     /root/reference (the reference globs its YAML configs and asset paths relative to the cwd), and populates it with the
     seeded synthetic SMPL model + checkpoints from glamr_amd/utils/synth.py,
  3. chdirs there and imports the reference packages.

It exists to (a) generate the golden fixtures under tests/golden/ (oracle/make_golden.py) and (b) pin oracle/port against
the real reference in the container-only tests.  /root/reference does not exist on the GPU box: nothing that runs there
may import this module.
"""
import os
import sys

REFERENCE_ROOT = '/root/reference'
_STATE = {}


def available():
    return os.path.isdir(os.path.join(REFERENCE_ROOT, 'global_recon'))


def setup(workdir='/tmp/glamr_ref_work', smpl_seed=1234, ckpt_seed=1):
    """Idempotent.  Returns the scratch directory (which becomes the process cwd)."""
    if _STATE.get('workdir') == workdir:
        return workdir
    if not available():
        raise RuntimeError('reference tree not present at %s' % REFERENCE_ROOT)
    repo_root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for p in (repo_root, os.path.join(repo_root, 'oracle', 'refstubs')):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.makedirs(workdir, exist_ok=True)
    for pkg in ('global_recon', 'lib', 'motion_infiller', 'traj_pred'):
        link = os.path.join(workdir, pkg)
        if not os.path.islink(link):
            os.symlink(os.path.join(REFERENCE_ROOT, pkg), link)
    from glamr_amd.utils import synth
    if not os.path.exists(os.path.join(workdir, 'data', 'J_regressor_extra.npy')):
        synth.write_smpl_assets(workdir, smpl_seed)
    os.chdir(workdir)
    if workdir not in sys.path:
        sys.path.insert(0, workdir)
    ck = os.path.join(workdir, 'results', 'traj_pred', 'traj_pred_demo', 'version_0', 'checkpoints', 'model-best-epoch=0000.ckpt')
    if not os.path.exists(ck):
        from glamr_amd.models.layouts import INFILLER_LAYOUT, TRAJPRED_LAYOUT
        synth.write_checkpoints(workdir, INFILLER_LAYOUT, TRAJPRED_LAYOUT, ckpt_seed)
    _STATE['workdir'] = workdir
    return workdir


def reference_optimizer(cfg_id='glamr_dynamic', device='cpu', log=None, out_dir=None):
    """GlobalReconOptimizer of the reference, constructed exactly as global_recon/run_demo.py:35,59 does."""
    import torch
    wd = setup()
    from global_recon.utils.config import Config
    from global_recon.models import model_dict
    cfg = Config(cfg_id, out_dir=out_dir or os.path.join(wd, 'out', cfg_id))
    model = model_dict[cfg.grecon_model_name](cfg, torch.device(device), log)
    return model, cfg


class QuietLog:
    """Drop-in for the reference logger that discards the per-iteration line (global_recon_model.py:646-659)."""

    def info(self, *a, **k):
        pass
