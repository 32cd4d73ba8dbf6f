"""DEVELOPMENT AID (GPU, under rocprofv3 --kernel-trace --stats): the latent-optimisation mode on one 300-frame sequence, K iterations of the first
stage with the iteration graph off (GLAMR_LATENT_GRAPH=0), so that the kernel statistics count the launches of an iteration by name."""
import os, sys
os.environ['GLAMR_LATENT_GRAPH'] = '0'
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import ensure_assets, CFG_ID, NUM_FRAMES
from glamr_amd.global_recon.configs import get_config
from glamr_amd.global_recon.models import model_dict
from glamr_amd.lib.models.smpl import SMPL
from glamr_amd.models.prior_models import MotionTrajJointModel
from glamr_amd.utils import synth
root = ensure_assets(); dev = torch.device('cuda:0')
smpl = SMPL(os.path.join(root, 'data', 'body_models', 'smpl'), pose_type='body26fk', extra_regressor_path=os.path.join(root, 'data', 'J_regressor_extra.npy')).to(dev)
mt = MotionTrajJointModel(None, dev, None, smpl=smpl, results_root=os.path.join(root, 'results'))
cfg = get_config(CFG_ID)
cfg['grecon_model_specs'].update(flag_opt_motion_latent=True, flag_opt_traj_latent=True)
ml = model_dict['global_recon_model'](cfg, dev, None, smpl=smpl, mt_model=mt)
d = synth.make_in_dict(seed=0, num_frames=NUM_FRAMES, num_persons=1, smpl_model=synth.make_smpl_model())
K = int(sys.argv[1]) if len(sys.argv) > 1 else 20
ml.optimize_batch([d], max_iters=K)
torch.cuda.synchronize()
print('done', K)
