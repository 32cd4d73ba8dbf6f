"""Evaluator on an MI355X against fixtures produced by the reference's global_recon/utils/evaluator.py
(oracle/make_golden.py gen_eval): same estimate, same ground truth, same metric names."""
import os

import numpy as np
import pytest
import torch

from oracle import make_golden as mg
from glamr_amd.utils import synth

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('cfg_id,T,P,dataset', mg.EVAL_CASES)
def test_metrics_match_the_reference_evaluator(asset_root, golden, cfg_id, T, P, dataset):
    from glamr_amd.global_recon.utils.evaluator import Evaluator
    from glamr_amd.lib.models.smpl import SMPL
    g = golden('eval_%s_T%d_P%d' % (cfg_id, T, P))
    dev = torch.device('cuda:0')
    md = synth.make_smpl_model()
    smpl = SMPL(os.path.join(asset_root, 'data', 'body_models', 'smpl'), pose_type='body26fk',
                extra_regressor_path=os.path.join(asset_root, 'data', 'J_regressor_extra.npy')).to(dev)
    ev = Evaluator(algo='glamr_amd', dataset=dataset, device=dev, align_freq=250, smpl=smpl, j_regressor_h36m=synth.make_h36m_regressor(md))
    data = {'seq_len': T, 'person_data': {}, 'gt': {}}
    for idx in range(P):
        data['person_data'][idx] = {k: g['in_p%d_%s' % (idx, k)] for k in ('smpl_orient_world', 'root_trans_world', 'smpl_pose', 'smpl_beta',
                                                                           'visible_orig', 'exist_frames')}
        data['person_data'][idx]['scale'] = None
        data['gt'][idx] = {k: g['in_gt%d_%s' % (idx, k)] for k in ('pose', 'shape', 'root_trans')}
    res = ev.compute_sequence_metrics(data, name='seq')
    for idx in range(P):
        pd = data['person_data'][idx]
        for k, tol in (('eval_joints_world', 2e-5), ('aligned_eval_joints_world', 2e-4), ('eval_joints_world_PA', 2e-4), ('aligned_trans', 1e-4)):
            err = np.abs(pd[k] - g['p%d_%s' % (idx, k)]).max()
            assert err < tol, '%s: %g' % (k, err)
        assert np.abs(data['gt'][idx]['eval_joints_world'] - g['gt%d_eval_joints_world' % idx]).max() < 2e-5
    for name, meter in res['metrics'].items():
        ref = g['metric_' + name]
        assert int(meter.count) == int(g['count_' + name]), name
        assert np.allclose(np.asarray(meter.avg), ref, rtol=2e-4, atol=2e-3), '%s: %s vs %s' % (name, meter.avg, ref)      # millimetres
    line = ev.print_metrics(print_accum=False)
    assert 'G-MPJPE' in line and 'PA-MPJPE-invis' in line
    # several seeds: the occluded-frame metrics take the best seed, the others the mean (evaluator.py:352-379)
    multi = ev.metrics_from_multiple_seeds([res, res])
    assert abs(multi['metrics']['G-MPJPE'].avg - res['metrics']['G-MPJPE'].avg) < 1e-6
