"""MI355X parity of the motion-prior kernels (glamr_nets_infer through the C ABI): infiller + trajectory predictor outputs vs
fixtures produced by the UNMODIFIED reference with the same supplied latents.  Tolerance 1e-4 (BASELINE.json / SURVEY.md 8c)."""
import glob
import os
import numpy as np
import pytest
import torch

from oracle import make_golden as mg
from oracle.port import build

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def priors(asset_root):
    from glamr_amd.models.priors import MotionPriorsHandle
    from glamr_amd.utils import synth
    sd = {}
    for name, sub in (('inf', 'motion_filler/motion_infiller_demo'), ('trj', 'traj_pred/traj_pred_demo')):
        path = sorted(glob.glob(os.path.join(asset_root, 'results', sub, 'version_*', 'checkpoints', '*best*.ckpt')))[-1]
        sd[name] = torch.load(path, map_location='cpu', weights_only=False)['state_dict']
    md = synth.make_smpl_model()
    rest = (md['J_regressor'].astype(np.float64) @ md['v_template'].astype(np.float64)).astype(np.float32)
    return MotionPriorsHandle(sd['inf'], sd['trj'], rest, synth.SMPL_PARENTS, torch.device('cuda:0'))


def _err(a, b):
    return float(np.abs(np.asarray(a, dtype=np.float64) - np.asarray(b, dtype=np.float64)).max())


@pytest.mark.parametrize('T', [120, 300])
def test_joint_inference_matches_reference(priors, golden, T):
    g = golden('nets')
    dev = torch.device('cuda:0')
    b = {k: torch.tensor(v, device=dev) for k, v in mg.net_inputs(T).items()}
    out = priors.infer(b['in_body_pose'], b['frame_mask'], [T], motion_eps=b['in_motion_latent'][None], traj_eps=b['in_traj_latent'])
    assert _err(out['pose'][0].cpu(), g['T%d_body_pose' % T][0, 0]) < 1e-4
    assert _err(out['local_traj'][0].cpu(), g['T%d_local_traj' % T][:, 0, 0]) < 1e-4
    assert _err(out['trans'][0].cpu(), g['T%d_trans' % T][0, 0]) < 2e-4
    assert _err(out['orient'][0].cpu(), g['T%d_orient' % T][0, 0]) < 2e-4


def test_single_padded_window(priors, golden):
    g = golden('nets')
    dev = torch.device('cuda:0')
    b = {k: torch.tensor(v, device=dev) for k, v in mg.net_inputs(40).items()}
    out = priors.infer(b['in_body_pose'], b['frame_mask'], [40], motion_eps=b['in_motion_latent'][None], traj_eps=None, traj=False)
    assert _err(out['pose'][0].cpu(), g['T40_body_pose'][0, 0]) < 1e-4


def test_ragged_batch_equals_single_runs(priors):
    """Sequences of different lengths in one padded batch must give what each gives alone (padding never leaks)."""
    dev = torch.device('cuda:0')
    lens = [300, 47, 120, 211]
    T = max(lens)
    ins = [mg.net_inputs(n, seed=i) for i, n in enumerate(lens)]
    pose = torch.zeros(len(lens), T, 69, device=dev)
    vis = torch.zeros(len(lens), T, device=dev)
    nw = max(x['in_motion_latent'].shape[0] for x in ins)
    meps = torch.zeros(len(lens), nw, 128, device=dev)
    teps = torch.zeros(len(lens), 128, device=dev)
    for i, x in enumerate(ins):
        pose[i, :lens[i]] = torch.tensor(x['in_body_pose'][0])
        vis[i, :lens[i]] = torch.tensor(x['frame_mask'][0]).float()
        meps[i, :x['in_motion_latent'].shape[0]] = torch.tensor(x['in_motion_latent'])
        teps[i] = torch.tensor(x['in_traj_latent'][0])
    batch = priors.infer(pose, vis, lens, motion_eps=meps, traj_eps=teps)
    for i, x in enumerate(ins):
        n = lens[i]
        single = priors.infer(pose[i:i + 1, :n].contiguous(), vis[i:i + 1, :n].contiguous(), [n], motion_eps=meps[i:i + 1], traj_eps=teps[i:i + 1])
        for key in ('pose', 'local_traj', 'trans', 'orient'):
            assert _err(batch[key][i, :n].cpu(), single[key][0].cpu()) < 1e-5, (i, key)
            assert float(batch[key][i, n:].abs().max()) == 0.0 if n < T else True


def test_large_batch_recurrence_equals_small_batch(priors):
    """Batches of 512+ sequences run the bi-LSTM on the matrix cores (16 sequences per workgroup, lstm_mfma_kernel); smaller ones
    one sequence per workgroup.  Same numbers up to the summation order of the recurrent product."""
    dev = torch.device('cuda:0')
    base = [97, 47, 120, 64, 11, 100]
    B = 523                                          # not a multiple of 16: the last workgroup is partly empty
    lens = [base[i % len(base)] for i in range(B)]
    T = max(lens)
    ins = [mg.net_inputs(n, seed=i) for i, n in enumerate(base)]
    pose = torch.zeros(B, T, 69, device=dev)
    vis = torch.zeros(B, T, device=dev)
    nw = max(x['in_motion_latent'].shape[0] for x in ins)
    meps = torch.zeros(B, nw, 128, device=dev)
    teps = torch.zeros(B, 128, device=dev)
    for i in range(B):
        x = ins[i % len(base)]
        pose[i, :lens[i]] = torch.tensor(x['in_body_pose'][0])
        vis[i, :lens[i]] = torch.tensor(x['frame_mask'][0]).float()
        meps[i, :x['in_motion_latent'].shape[0]] = torch.tensor(x['in_motion_latent'])
        teps[i] = torch.tensor(x['in_traj_latent'][0])
    big = priors.infer(pose, vis, lens, motion_eps=meps, traj_eps=teps)
    small = priors.infer(pose[:len(base)].contiguous(), vis[:len(base)].contiguous(), lens[:len(base)], motion_eps=meps[:len(base)].contiguous(),
                         traj_eps=teps[:len(base)].contiguous())
    for i in (0, 1, 5, 7, 300, 517, 522):
        j, n = i % len(base), lens[i]
        for key in ('local_traj', 'trans', 'orient', 'pose'):
            assert _err(big[key][i, :n].cpu(), small[key][j, :n].cpu()) < 2e-5, (i, key)
