"""MI355X: the per-iteration loss log of the optimiser stage (glamr_scene_batch.loss_history through the C ABI, GlobalReconOptimizer(log=...)).

The reference calls write_logs after every optimizer.step with the UNWEIGHTED value of every term of the stage's loss_cfg
(global_recon/models/global_recon_model.py:564, 646-659).  Here a stage is one kernel launch; a launch that is given a history array records
those values for every iteration.  Checked: every row against the CPU restatement of the reference's loop (oracle/port, its own log hook),
row 0 against the first-iteration losses of the unmodified reference (tests/golden/grecon_*.npz), the last row against `losses`, the
optimisation result against a launch without the history, and the lines a `log` object receives against the reference's line format."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

from oracle import make_golden as mg
from oracle.port import build
from glamr_amd import _lib
from glamr_amd.global_recon import packing
from glamr_amd.global_recon.configs import get_config
from glamr_amd.utils import synth
from tests.grecon_common import j_local_from_oracle

pytestmark = pytest.mark.gpu


def _run(packed, sd, dev):
    L = _lib.lib()
    sb = packed.struct()
    ws = torch.empty(L.glamr_grecon_workspace_bytes(packed.S, packed.P, packed.T), dtype=torch.uint8, device=dev)
    _lib.check(L.glamr_grecon_run_stage(ctypes.byref(sb), ctypes.byref(sd), None, _lib.ptr(ws), _lib.current_stream()))
    torch.cuda.synchronize()


@pytest.mark.parametrize('cfg_id,T,P,K', [('glamr_dynamic', 120, 1, 8), ('glamr_static_multi', 120, 2, 6)])
def test_loss_history_of_a_stage_launch(asset_root, golden, cfg_id, T, P, K):
    dev = torch.device('cuda:0')
    g = golden('grecon_%s_T%d_P%d' % (cfg_id, T, P))
    cfg = get_config(cfg_id)
    specs = cfg['grecon_model_specs']
    in_dict = synth.make_in_dict(seed=3, num_frames=T, num_persons=P, smpl_model=synth.make_smpl_model())
    rows = {}
    ora = build.load_optimizer(asset_root, cfg, log_fn=lambda stage, it, uw: rows.setdefault(stage, []).append(uw))
    data = ora.init_data(in_dict, latents=mg.latents_for(in_dict, 3))
    jl = j_local_from_oracle(ora.smpl, data)
    stage, spec = next(iter(cfg['opt_stage_specs'].items()))
    sd = packing.stage_desc(spec, specs, False, niters=K)
    with_h = packing.PackedScenes([data], [jl], dev)
    with_h.t['loss_history'] = torch.full((1, K, len(packing.LOSS_IDS)), float('nan'), device=dev)
    _run(with_h, sd, dev)
    plain = packing.PackedScenes([data], [jl], dev)
    _run(plain, sd, dev)
    h = with_h.t['loss_history'][0].cpu().numpy()
    names = [n for n in spec['loss_cfg'] if n in packing.LOSS_IDS]
    assert np.isfinite(h[:, [packing.LOSS_IDS[n] for n in names]]).all()
    # row 0: the first-iteration values of the UNMODIFIED reference
    for n in names:
        key = '%s_loss_%s' % (stage, n)
        if key in g:
            ref = float(g[key])
            assert abs(h[0, packing.LOSS_IDS[n]] - ref) <= 2e-4 * max(1.0, abs(ref)), (n, h[0, packing.LOSS_IDS[n]], ref)
    # every row: the restated reference loop's own log (the trajectories are free-running: bounds as for the K-step states)
    ora.optimize_main(data, spec['opt_variables'], spec['opt_lr'], K, spec['loss_cfg'], {'stage': stage})
    worst = 0.0
    for it in range(K):
        for n in names:
            ref = rows[stage][it][n]
            err = abs(h[it, packing.LOSS_IDS[n]] - ref) / max(1.0, abs(ref))
            worst = max(worst, err)
    print('loss history %s, %d iterations x %d terms: worst relative difference from the restated reference loop %.2e' % (cfg_id, K, len(names), worst))
    assert worst < 2e-3
    # the last row is the launch's `losses`; the optimisation itself is the one a launch without history makes
    assert np.array_equal(h[K - 1], with_h.t['losses'][0].cpu().numpy())
    kp_a, kp_b = with_h.t['kp_2d_pred'].cpu().numpy(), plain.t['kp_2d_pred'].cpu().numpy()
    assert float(np.abs(kp_a - kp_b).max()) < 5e-3
    # (another instance of the same algorithm: parameters whose gradient is structurally zero take a +-lr first step whose sign is rounding
    # noise in either -- the same bound the K-step states of tests/grecon_common.py live with)
    assert float((with_h.t['params'] - plain.t['params']).abs().max().cpu()) < 5e-3


def test_log_object_receives_the_reference_lines(asset_root):
    from glamr_amd.global_recon.models import model_dict
    from glamr_amd.lib.models.smpl import SMPL
    from glamr_amd.models.prior_models import MotionTrajJointModel
    dev = torch.device('cuda:0')

    class Log:
        def __init__(self):
            self.lines = []

        def info(self, msg, *a, **k):
            self.lines.append(str(msg))

    log = Log()
    smpl = SMPL(os.path.join(asset_root, 'data', 'body_models', 'smpl'), pose_type='body26fk',
                extra_regressor_path=os.path.join(asset_root, 'data', 'J_regressor_extra.npy')).to(dev)
    mt = MotionTrajJointModel(None, dev, None, smpl=smpl, results_root=os.path.join(asset_root, 'results'))
    cfg = get_config('glamr_static_multi')                        # two stages
    model = model_dict['global_recon_model'](cfg, dev, log, smpl=smpl, mt_model=mt)
    in_dict = synth.make_in_dict(seed=5, num_frames=60, num_persons=2, smpl_model=synth.make_smpl_model(), seq_name='clip7')
    K = 4
    out = model.optimize(in_dict, latents=mg.latents_for(in_dict, 5), max_iters=K)
    stage_lines = [l for l in log.lines if ' | LR: ' in l]
    assert len(stage_lines) == K * len(cfg['opt_stage_specs'])
    # f'{cfg.id} - {seq_name} - {stage} | {cur_iter:4d}/{opt_niters} | TE: {..} ETA: {..} | LR: {opt_lr:.0e} | name: {value:7.3f} | ...'  (:654-655)
    pat = re.compile(r'^glamr_static_multi - clip7 - (\w+) \| +(\d+)/%d \| TE: \d+:\d\d:\d\d ETA: \d+:\d\d:\d\d \| LR: \de[-+]\d\d \| (.+)$' % K)
    for (stage, spec), chunk in zip(cfg['opt_stage_specs'].items(), [stage_lines[i * K:(i + 1) * K] for i in range(len(cfg['opt_stage_specs']))]):
        for it, line in enumerate(chunk):
            m = pat.match(line)
            assert m, line
            assert m.group(1) == stage and int(m.group(2)) == it
            terms = [t.split(': ') for t in m.group(3).split(' | ')]
            assert [t[0] for t in terms] == [n for n in spec['loss_cfg'] if n in packing.LOSS_IDS]
            assert all(np.isfinite(float(t[1])) for t in terms)
        assert model.loss_history[stage].shape == (1, K, len(packing.LOSS_IDS))
    assert 'person_data' in out
