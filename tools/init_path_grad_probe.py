import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bench import ensure_assets
from oracle import make_golden as mg
from glamr_amd import _lib
from glamr_amd.utils import synth
from glamr_amd.global_recon import packing
from glamr_amd.global_recon.models import model_dict
from glamr_amd.global_recon.configs import get_config
from glamr_amd.lib.models.smpl import SMPL
from glamr_amd.models.prior_models import MotionTrajJointModel
root = ensure_assets(); dev = torch.device('cuda:0')
smpl = SMPL(os.path.join(root, 'data', 'body_models', 'smpl'), pose_type='body26fk', extra_regressor_path=os.path.join(root, 'data', 'J_regressor_extra.npy')).to(dev)
mt = MotionTrajJointModel(None, dev, None, smpl=smpl, results_root=os.path.join(root, 'results'))
in_dict = synth.make_in_dict(seed=0, num_frames=300, num_persons=1, smpl_model=synth.make_smpl_model())
lat = mg.latents_for(in_dict, 0)
m = model_dict['global_recon_model'](get_config('glamr_dynamic'), dev, None, smpl=smpl, mt_model=mt)
_, pa = m.init_data_batch([in_dict], [lat])
_, pb = m.init_data_batch_host([in_dict], [lat])
L = _lib.lib()
np.set_printoptions(linewidth=220, precision=6)
has_wd = False
for stage, spec in m.opt_stage_specs.items():
    print('=== stage', stage, spec['opt_variables'])
    for it in range(3):
        res = []
        for p in (pa, pb):
            sd = packing.stage_desc(spec, m.specs, has_world_dheading=has_wd, niters=1)
            sb = p.struct()
            grads = torch.zeros_like(p.t['params'])
            ws = torch.empty(L.glamr_grecon_workspace_bytes(p.S, p.P, p.T), dtype=torch.uint8, device=dev)
            _lib.check(L.glamr_grecon_run_stage(ctypes.byref(sb), ctypes.byref(sd), _lib.ptr(grads), _lib.ptr(ws), _lib.current_stream()))
            torch.cuda.synchronize()
            res.append((grads.cpu().numpy()[0], p.t['params'].cpu().numpy()[0], p.t['cam_pose'].cpu().numpy()[0]))
        l, T = pa.layout, pa.T
        (ga, qa, ca), (gb, qb, cb) = res
        if 'cam' in spec['opt_variables']:
            for nm, w in (('cam_rot6d', 6), ('cam_trans', 3)):
                A = ga[l[nm]:l[nm] + w * T].reshape(T, w); B = gb[l[nm]:l[nm] + w * T].reshape(T, w)
                e = np.abs(A - B) / (np.abs(B) + 1e-12)
                print(' it', it, nm, 'grad frames with rel diff > 1e-2:', np.where(e.max(1) > 1e-2)[0][:20])
                for t in ():
                    print('     t', t, 'dev', A[t], 'host', B[t])
        d = np.abs(qa - qb)
        bad = np.where(d > 1e-4)[0]
        print(' it', it, 'n differing params', len(bad))
        for i in bad[:40]:
            print('      idx', i, 'rel', i - l['person0'] if i >= l['person0'] else i, 'grad dev %.6e host %.6e  param dev %.6e host %.6e' % (ga[i], gb[i], qa[i], qb[i]))
        print(' it', it, 'param max diff', d.max(), 'at', d.argmax(), {k: v for k, v in l.items() if isinstance(v, int)} if it == 0 and stage == list(m.opt_stage_specs)[0] else '')
    has_wd = has_wd or 'world_dheading' in spec['opt_variables']
    break
