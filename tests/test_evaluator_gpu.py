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


def test_device_alignment_kernels_match_the_numpy_twins(asset_root):
    """glamr_eval_procrustes / glamr_eval_heading_align / glamr_eval_regress_joints against the numpy restatements of
    lib/utils/torch_transform.py:282-345, traj_pred/utils/traj_utils.py:97-107 and the regression matmul -- including the cases an SVD routine
    disagrees with itself on: a mirrored target (det < 0: the reflection fix), planar joints (a zero singular value), a chunk boundary."""
    from glamr_amd.global_recon.utils import evaluator as E
    from glamr_amd.lib.models.smpl import SMPL
    from glamr_amd.lib.utils import np_transform as nt
    dev = torch.device('cuda:0')
    md = synth.make_smpl_model()
    smpl = SMPL(os.path.join(asset_root, 'data', 'body_models', 'smpl'), pose_type='body26fk',
                extra_regressor_path=os.path.join(asset_root, 'data', 'J_regressor_extra.npy')).to(dev)
    ev = E.Evaluator(algo='t', dataset='', device=dev, align_freq=250, smpl=smpl, j_regressor_h36m=synth.make_h36m_regressor(md))
    rng = np.random.default_rng(5)
    N, J = 700, 14
    S1 = rng.normal(size=(N, J, 3)).astype(np.float32) * 0.3
    Rr = nt.aa_to_rotmat(rng.normal(size=(N, 3)).astype(np.float32))
    S2 = (1.3 * np.einsum('nab,njb->nja', Rr, S1) + rng.normal(size=(N, 1, 3)) + 0.01 * rng.normal(size=(N, J, 3))).astype(np.float32)
    S2[100:200, :, 0] *= -1.0                      # mirrored targets: the optimal ROTATION needs the sign fix of the last singular vector
    S1[300:350, :, 2] = 0.0                        # planar source: K has a zero singular value
    S2[300:350] = S1[300:350] * 0.7 + 0.2
    got = ev.procrustes(S1, S2)
    ref = E.batch_compute_similarity_transform(S1, S2)
    assert np.abs(got - ref).max() < 2e-5, np.abs(got - ref).max()
    # a noise-free similarity transform is recovered exactly
    target = (1.3 * np.einsum('nab,njb->nja', Rr[:50], S1[:50]) + 0.5).astype(np.float32)
    assert np.abs(ev.procrustes(S1[:50], target) - target).max() < 2e-5
    # heading alignment over three chunks (250 + 250 + 113 frames, one frame of overlap each)
    n = 613
    t = np.arange(n)[:, None] / 30.0
    aa = np.concatenate([np.full((n, 1), np.pi / 2) + 0.1 * np.sin(t), 0.2 * np.cos(0.7 * t), 0.5 * t + 0.3 * np.sin(1.3 * t)], axis=1).astype(np.float32)
    tr = np.concatenate([np.cos(0.4 * t) * 2, t * 0.8, 0.9 + 0.05 * np.sin(3 * t)], axis=1).astype(np.float32)
    pd = {'smpl_orient_world': aa, 'root_trans_world': tr}
    ev.get_aligned_orient_trans(pd)
    q_all = nt.aa_to_quat(aa)
    qs, ts = [], []
    for i in range(int(np.ceil(n / 250))):
        sind, eind = i * 250 - int(i > 0), min((i + 1) * 250, n)
        q, x = E.convert_traj_world2heading(q_all[sind:eind], tr[sind:eind], apply_base_orient_after=True)
        qs.append(q[int(i > 0):]); ts.append(x[int(i > 0):])
    ref_q, ref_t = np.concatenate(qs), np.concatenate(ts)
    assert np.abs(pd['aligned_trans'] - ref_t).max() < 1e-5
    assert np.minimum(np.abs(pd['aligned_orient_q'] - ref_q).max(axis=1), np.abs(pd['aligned_orient_q'] + ref_q).max(axis=1)).max() < 1e-5
    assert np.abs(nt.aa_to_rotmat(pd['aligned_orient']) - nt.quat_to_rotmat(ref_q)).max() < 2e-5
    # regression: (17, V) x (B, V, 3)
    verts = torch.as_tensor(rng.normal(size=(9, md['v_template'].shape[0], 3)).astype(np.float32), device=dev)
    j17 = torch.empty((9, 17, 3), device=dev)
    from glamr_amd import _lib
    _lib.check(_lib.lib().glamr_eval_regress_joints(9, verts.shape[1], 17, _lib.ptr(verts), _lib.ptr(ev.J_regressor), _lib.ptr(j17), _lib.current_stream()))
    ref_j = np.einsum('jv,bvc->bjc', ev.J_regressor.cpu().numpy().astype(np.float64), verts.cpu().numpy().astype(np.float64))
    assert np.abs(j17.cpu().numpy() - ref_j).max() < 2e-5
