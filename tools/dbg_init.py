import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bench import ensure_assets
from oracle import make_golden as mg
from glamr_amd.utils import synth
from glamr_amd.global_recon.models import model_dict
from glamr_amd.global_recon.configs import get_config
from glamr_amd.lib.models.smpl import SMPL
from glamr_amd.models.prior_models import MotionTrajJointModel
root = ensure_assets(); dev = torch.device('cuda:0')
smpl = SMPL(os.path.join(root, 'data', 'body_models', 'smpl'), pose_type='body26fk', extra_regressor_path=os.path.join(root, 'data', 'J_regressor_extra.npy')).to(dev)
mt = MotionTrajJointModel(None, dev, None, smpl=smpl, results_root=os.path.join(root, 'results'))
cfg_id, T, P, seed = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
in_dict = synth.make_in_dict(seed=seed, num_frames=T, num_persons=P, smpl_model=synth.make_smpl_model())
m = model_dict['global_recon_model'](get_config(cfg_id), dev, None, smpl=smpl, mt_model=mt)
lat = mg.latents_for(in_dict, seed)
a = m.init_data(in_dict, latents=lat)
m.init_data_batch = m.init_data_batch_host
b = m.init_data(in_dict, latents=lat)
def cmp(x, y, path=''):
    if isinstance(x, dict):
        for k in x:
            if k in y: cmp(x[k], y[k], path + '/' + str(k))
    elif isinstance(x, np.ndarray) and isinstance(y, np.ndarray) and x.shape == y.shape and x.dtype != object:
        e = np.abs(x.astype(np.float64) - y.astype(np.float64))
        if e.size and e.max() > 1e-5: print('%-40s %.3e at %s' % (path, e.max(), np.unravel_index(e.argmax(), e.shape)))
    elif isinstance(x, np.ndarray) and isinstance(y, np.ndarray): print('shape', path, x.shape, y.shape)
cmp(a, b)
print('---- packed tensors')
m2 = model_dict['global_recon_model'](get_config(cfg_id), dev, None, smpl=smpl, mt_model=mt)
_, pa = m2.init_data_batch([in_dict], [lat])
_, pb = m2.init_data_batch_host([in_dict], [lat])
for k in pa.t:
    x, y = pa.t[k], pb.t.get(k)
    if y is None or x is None or x.shape != y.shape: print('skip', k, None if x is None else tuple(x.shape), None if y is None else tuple(y.shape)); continue
    e = (x.double() - y.double()).abs()
    if e.numel() and e.max() > 1e-6: print('%-20s %.3e at %s' % (k, e.max().item(), np.unravel_index(int(e.argmax()), tuple(e.shape))))
    sx, sy = torch.signbit(x.float()), torch.signbit(y.float())
    if (sx != sy).any() and k in ('cam_pose',): print('  signbit differences in', k, int((sx != sy).sum()))
