"""DEVELOPMENT AID (GPU): latent-optimisation mode, ms per iteration for one 300-frame sequence (slope between a 12- and a 52-iteration run, as
bench.py's `latent_optimisation_mode`), and the latency of one sequence through the plain mode.  usage: python tools/latent_time.py [label]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from glamr_amd.global_recon.configs import get_config
from glamr_amd.global_recon.models import model_dict
from glamr_amd.utils import synth
dev = torch.device('cuda:0')
model = bench.build_model(bench.ensure_assets(), dev)
md = synth.make_smpl_model()
cfg = get_config(bench.CFG_ID)
cfg['grecon_model_specs'].update(flag_opt_motion_latent=True, flag_opt_traj_latent=True)
ml = model_dict['global_recon_model'](cfg, dev, None, smpl=model.smpl, mt_model=model.mt_model)
one = synth.make_in_dict(seed=0, num_frames=bench.NUM_FRAMES, num_persons=1, smpl_model=md)
ml.optimize(one, max_iters=3)


def run_k(k):
    torch.cuda.synchronize()
    t0 = time.time()
    ml.optimize(one, max_iters=k)
    torch.cuda.synchronize()
    return time.time() - t0


K1, K2 = 12, 52
t1, t2 = min(run_k(K1), run_k(K1)), min(run_k(K2), run_k(K2))
lat = bench.one_sequence_latency(model, one)
print('%-10s latent mode %.2f ms per iteration | one sequence, plain mode: %s' % (sys.argv[1] if len(sys.argv) > 1 else '', (t2 - t1) * 1e3 / (K2 - K1),
                                                                                 {k: lat[k] for k in ('host_dict_to_host_dict', 'hbm_to_hbm')}))
