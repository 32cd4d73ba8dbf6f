"""TEST INFRASTRUCTURE ONLY -- assembles the CPU port from an asset directory (synthetic or real):
<root>/data/body_models/smpl/SMPL_NEUTRAL.pkl, <root>/data/J_regressor_extra.npy and the two checkpoints under
<root>/results/... (paths as hard-wired in the reference: lib/models/smpl.py:28-31, motion_traj_joint_model.py:37-65)."""
import glob
import os
import torch
from .smpl import SMPL
from .nets import MotionInfillerVAE, TrajPredVAE, MotionTrajJointModel
from .grecon import GlobalReconOptimizer


def ensure_synthetic_assets(root, smpl_seed=1234, ckpt_seed=1):
    from glamr_amd.utils import synth
    from glamr_amd.models.layouts import INFILLER_LAYOUT, TRAJPRED_LAYOUT
    if not os.path.exists(os.path.join(root, 'data', 'J_regressor_extra.npy')):
        synth.write_smpl_assets(root, smpl_seed)
    if not glob.glob(os.path.join(root, 'results', 'traj_pred', 'traj_pred_demo', 'version_0', 'checkpoints', '*best*.ckpt')):
        synth.write_checkpoints(root, INFILLER_LAYOUT, TRAJPRED_LAYOUT, ckpt_seed)
    return root


def _ckpt(root, sub):
    return sorted(glob.glob(os.path.join(root, 'results', sub, 'version_*', 'checkpoints', '*best*.ckpt')))[-1]


def load_smpl(root):
    return SMPL(os.path.join(root, 'data', 'body_models', 'smpl'), pose_type='body26fk',
                extra_regressor_path=os.path.join(root, 'data', 'J_regressor_extra.npy'))


def load_joint_model(root, smpl=None, device=torch.device('cpu')):
    smpl = smpl if smpl is not None else load_smpl(root)
    inf, trj = MotionInfillerVAE(), TrajPredVAE(smpl=smpl)
    for net, sub in ((inf, 'motion_filler/motion_infiller_demo'), (trj, 'traj_pred/traj_pred_demo')):
        sd = torch.load(_ckpt(root, sub), map_location='cpu', weights_only=False)['state_dict']
        net.load_state_dict({k: v for k, v in sd.items() if not k.startswith('smpl.')}, strict=True)
    return MotionTrajJointModel(inf, trj, device)


def load_optimizer(root, cfg_dict, device=torch.device('cpu'), log_fn=None):
    smpl = load_smpl(root).to(device)
    return GlobalReconOptimizer(cfg_dict, smpl, load_joint_model(root, smpl, device), device, log_fn)
