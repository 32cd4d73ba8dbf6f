"""Development aid: bit-level fingerprint of the device init_data (per-person preparation, priors, scene assembly, cached joints, first forward pass)
over a set of synthetic inputs, for A/B runs of two builds of the library (GLAMR_LIB_PATH).  A change that must not move a bit of the
initial state -- which decides the basin of a sequence with a detection gap -- keeps every fingerprint."""
import hashlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from glamr_amd.global_recon.configs import get_config
from glamr_amd.global_recon.models import model_dict
from glamr_amd.lib.models.smpl import SMPL
from glamr_amd.models.prior_models import MotionTrajJointModel
from glamr_amd.utils import synth
from oracle.port import build
from oracle import make_golden as mg

dev = torch.device('cuda:0')
root = build.ensure_synthetic_assets('/tmp/glamr_bench_assets')
smpl = SMPL(os.path.join(root, 'data', 'body_models', 'smpl'), pose_type='body26fk', extra_regressor_path=os.path.join(root, 'data', 'J_regressor_extra.npy')).to(dev)
mt = MotionTrajJointModel(None, dev, None, smpl=smpl, results_root=os.path.join(root, 'results'))
md = synth.make_smpl_model()
cases = [('glamr_dynamic', 300, 1, s, None) for s in range(12)] + [('glamr_dynamic', 300, 1, s, (0, 0)) for s in (0, 1)] + \
        [('glamr_static_multi', 300, 4, 38, None), ('glamr_dynamic_multi', 90, 2, 3, None), ('glamr_3dpw', 120, 1, 3, None), ('glamr_h36m', 100, 2, 3, None)]
for cfg_id, T, P, seed, gap in cases:
    model = model_dict['global_recon_model'](get_config(cfg_id), dev, None, smpl=smpl, mt_model=mt)
    in_dict = synth.make_in_dict(seed=seed, num_frames=T, num_persons=P, smpl_model=md, gap=gap)
    rin = model.stage_inputs([in_dict], [mg.latents_for(in_dict, seed)])
    datas, packed = model.init_resident(rin)
    torch.cuda.synchronize()
    h = hashlib.sha1()
    for k in sorted(packed.t):
        v = packed.t[k]
        if torch.is_tensor(v):
            h.update(v.detach().cpu().numpy().tobytes())
    for k in sorted(packed.person_arrays):
        h.update(packed.person_arrays[k].detach().cpu().numpy().tobytes())
    print('init %-20s T=%d P=%d seed=%d gap=%s  %s' % (cfg_id, T, P, seed, gap, h.hexdigest()[:16]), flush=True)
