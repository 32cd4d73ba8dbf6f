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
in_dict = synth.make_in_dict(seed=0, num_frames=300, num_persons=1, smpl_model=synth.make_smpl_model())
lat = mg.latents_for(in_dict, 0)
m2 = model_dict['global_recon_model'](get_config('glamr_dynamic'), dev, None, smpl=smpl, mt_model=mt)
_, pa = m2.init_data_batch([in_dict], [lat])
_, pb = m2.init_data_batch_host([in_dict], [lat])
print('keys only in a', set(pa.t) - set(pb.t), 'only in b', set(pb.t) - set(pa.t))
print('T', pa.T, pb.T, 'P', pa.P, pb.P)
for k in pa.t:
    x, y = pa.t[k], pb.t.get(k)
    if x is None or y is None: print(k, 'None', x is None, y is None); continue
    print('%-22s %-18s %-18s %s %s nan %d/%d' % (k, tuple(x.shape), tuple(y.shape), x.dtype, y.dtype, int(torch.isnan(x.float()).sum()), int(torch.isnan(y.float()).sum())), end=' ')
    if x.shape == y.shape:
        e = (x.double() - y.double()).abs()
        sx, sy = torch.signbit(x.float()), torch.signbit(y.float())
        print('maxdiff %.2e signdiff %d contiguous %s %s' % (e.max().item() if e.numel() else 0, int((sx != sy).sum()), x.is_contiguous(), y.is_contiguous()))
    else: print()
