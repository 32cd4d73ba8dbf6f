"""DEVELOPMENT AID (build container, CPU): the kernel algorithm on the CPU runtime (tests/hostsim) from the ORACLE's initial state (= the
reference's, bit for bit) through the full 500-iteration schedule of BASELINE configs[1] WITH the detection gap, for several seeds, against
the multi-seed goldens (oracle/make_golden.py gen_full_seeds) and their 1e-6 families.  Predicts what the host-init leg of
tests/test_e2e_gpu.py::test_full_schedule_detection_gap_multi_seed finds on the MI355X.  python tools/seeds_hostsim.py 0 1 2 ..."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from oracle import make_golden as mg
from oracle.port import build
from glamr_amd.global_recon import packing
from glamr_amd.global_recon.configs import get_config
from glamr_amd.utils import synth
from tests import grecon_common as gc
import tempfile

seeds = [int(x) for x in sys.argv[1:]] or [0]
run, dev = gc.hostsim_runner()
cfg = get_config('glamr_dynamic')
root = build.ensure_synthetic_assets(os.path.join(tempfile.gettempdir(), 'glamr_test_assets'))
ora = build.load_optimizer(root, cfg)
md = synth.make_smpl_model()
for seed in seeds:
    name = 'full_glamr_dynamic_T300' if seed == 0 else mg.seed_name(seed)
    g = np.load(os.path.join(mg.GOLD, name + '.npz'))
    in_dict = synth.make_in_dict(seed=seed, num_frames=300, num_persons=1, smpl_model=md)
    data = ora.init_data(in_dict, latents=mg.latents_for(in_dict, seed))
    packed = packing.PackedScenes([data], [gc.j_local_from_oracle(ora.smpl, data)], dev)
    spec = cfg['opt_stage_specs']['init_opt']
    t0 = time.time()
    run(packed, packing.stage_desc(spec, cfg['grecon_model_specs'], False), False)
    vis = g['p0_vis_frames']
    kp = packed.t['kp_2d_pred'][0, :300].numpy()
    d = np.abs(kp - g['p0_kp_2d_pred'])[vis].max(axis=(1, 2))
    fam = {k[4:-len('_kp_2d_pred')]: float(np.abs(kp - g[k])[vis].max()) for k in g.files if k.startswith('fam_') and k.endswith('_kp_2d_pred')}
    print('seed %d: CPU runtime vs golden max %.4f px (frames > 1 px: %d); vs family %s; %.0f s' % (seed, d.max(), int((d > 1).sum()), {k: round(v, 3) for k, v in fam.items()}, time.time() - t0), flush=True)
