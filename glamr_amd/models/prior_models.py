"""Drop-in host classes of the two motion priors and their wrapper.  Same names, constructor arguments, `load_from_checkpoint`,
`.inference(...)` signatures and output dictionary keys as the reference:

    MotionInfillerVAE      motion_infiller/models/motion_infiller_vae.py:440-667
    TrajPredVAE            traj_pred/models/traj_pred_vae.py:341-548
    MotionTrajJointModel   motion_infiller/models/motion_traj_joint_model.py:17-145

The networks themselves run in HIP kernels (glamr_nets_* ABI); these classes hold the checkpoint tensors, draw the Gaussian
latents when the caller does not supply them (lib/utils/dist.py:21-23) and arrange outputs in the reference's layouts.
Training-time paths (`forward` = encoder + posterior + decoder in 'train' mode, `training_step`, recon) are not part of the
inference hot path and raise NotImplementedError.
"""
import glob
import os

import numpy as np
import torch

from .layouts import INFILLER_LAYOUT, TRAJPRED_LAYOUT
from .priors import MotionPriorsHandle, num_windows, NZ, PAST, CUR

FUT = 10
SMPL_PARENTS = [-1, 0, 0, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 9, 9, 12, 13, 14, 16, 17, 18, 19, 20, 21]


def _zeros_sd(layout):
    return {k: torch.zeros(tuple(s)) for k, s in layout}


class _PriorBase:
    LAYOUT = None

    def __init__(self, cfg=None):
        self.cfg = cfg
        self.nz = NZ
        self._sd = {k: torch.zeros(tuple(s)) for k, s in self.LAYOUT}
        self.device = torch.device('cpu')
        self.training = False
        self._handle = None

    # -- nn.Module-like surface the reference call sites use (motion_traj_joint_model.py:44-49,65-69) -----------------------
    @classmethod
    def load_from_checkpoint(cls, checkpoint_path, cfg=None, strict=True, map_location=None, **kwargs):
        ckpt = torch.load(checkpoint_path, map_location='cpu', weights_only=False)
        model = cls(cfg)
        model.load_state_dict(ckpt['state_dict'], strict=strict)
        return model

    def load_state_dict(self, sd, strict=True):
        missing = [k for k in self._sd if k not in sd]
        unexpected = [k for k in sd if k not in self._sd]
        if missing or (strict and unexpected):
            raise RuntimeError('checkpoint mismatch: missing %s unexpected %s' % (missing[:5], unexpected[:5] if strict else []))
        for k in self._sd:
            if tuple(sd[k].shape) != tuple(self._sd[k].shape):
                raise RuntimeError('size mismatch for %s' % k)
            self._sd[k] = sd[k].detach().float().cpu().clone()
        self._handle = None

    def state_dict(self):
        return dict(self._sd)

    def parameters(self):
        return iter(self._sd.values())

    def to(self, device):
        self.device = torch.device(device)
        return self

    def eval(self):
        self.training = False
        return self

    def forward(self, data):
        raise NotImplementedError('the training-mode forward (posterior encoder + decoder, %s) is outside the MI355X inference hot '
                                  'path; use .inference(...)' % type(self).__name__)

    __call__ = forward


class MotionInfillerVAE(_PriorBase):
    LAYOUT = INFILLER_LAYOUT
    model_type = 'angle'
    past_nframe, cur_nframe, fut_nframe = PAST, CUR, FUT

    def get_latent(self, seq_len):
        return torch.randn((num_windows(seq_len), self.nz))

    def _ensure_handle(self, rest_joints=None):
        if self._handle is None:
            self._handle = MotionPriorsHandle(self._sd, _zeros_sd(TRAJPRED_LAYOUT), np.zeros((24, 3), np.float32), SMPL_PARENTS, self.device)
        return self._handle

    def inference(self, batch, sample_num=5, recon=False, multi_step=False, _handle=None):
        """batch: {'in_body_pose' (B,T,69), 'frame_mask' (B,T) 1 = visible[, 'in_motion_latent' (n_windows,128)]}.
        Returns the reference's keys: infer_out_body_pose (B,S,T,69), infer_out_pose (B,S,T,72), in_body_pose_tp (T,B,69)."""
        if recon or not multi_step:
            raise NotImplementedError('only inference(recon=False, multi_step=True) -- the call GLAMR makes -- is implemented')
        h = _handle or self._ensure_handle()
        pose = batch['in_body_pose'].to(self.device).float()
        vis = (batch['frame_mask'].to(self.device) == 1).float()
        B, T = pose.shape[:2]
        nw = num_windows(T)
        outs = []
        for _ in range(sample_num):
            if 'in_motion_latent' in batch:
                eps = batch['in_motion_latent'].to(self.device).float().view(1, nw, NZ).expand(B, -1, -1).contiguous()
            else:
                eps = torch.randn((B, nw, NZ), device=self.device)
            outs.append(h.infer(pose, vis, [T] * B, motion_eps=eps, traj=False)['pose'])
        body = torch.stack(outs, dim=1)                                   # (B,S,T,69)
        data = dict(batch)
        data['infer_out_body_pose'] = body
        data['infer_out_pose'] = torch.cat([torch.zeros_like(body[..., :3]), body], dim=-1)
        data['in_body_pose_tp'] = outs[-1].transpose(0, 1).contiguous()   # the running input ends up holding the last sample
        data['batch_size'], data['seq_len'] = B, T
        return data


class TrajPredVAE(_PriorBase):
    LAYOUT = TRAJPRED_LAYOUT
    model_type = 'joint'
    in_joint_pos_only = False

    def __init__(self, cfg=None, rest_joints=None):
        super().__init__(cfg)
        self.rest_joints = rest_joints

    def get_latent(self, seq_len):
        return torch.zeros((1, self.nz))

    def _ensure_handle(self):
        if self._handle is None:
            if self.rest_joints is None:
                raise RuntimeError('TrajPredVAE needs the SMPL rest joints (J_regressor @ v_template) for its forward kinematics: '
                                   'pass rest_joints= or use MotionTrajJointModel')
            self._handle = MotionPriorsHandle(_zeros_sd(INFILLER_LAYOUT), self._sd, self.rest_joints, SMPL_PARENTS, self.device)
        return self._handle

    def inference(self, batch, sample_num=5, recon=False, recon_only=False, multi_step=False, _handle=None):
        """batch: {'in_body_pose' (B,T,69)[, 'in_traj_latent' (B,128)]}.  Returns infer_out_local_traj_tp (T,B,S,11),
        infer_out_trans / infer_out_orient (B,S,T,3), infer_out_pose (B,S,T,72)."""
        if recon or recon_only or multi_step:
            raise NotImplementedError('only inference(recon=False, multi_step=False) -- the call GLAMR makes -- is implemented')
        h = _handle or self._ensure_handle()
        pose = batch['in_body_pose'].to(self.device).float()
        B, T = pose.shape[:2]
        res = {k: [] for k in ('local_traj', 'trans', 'orient')}
        for _ in range(sample_num):
            eps = batch['in_traj_latent'].to(self.device).float().expand(B, -1).contiguous() if 'in_traj_latent' in batch else torch.randn((B, NZ), device=self.device)
            o = h.infer(pose, None, [T] * B, traj_eps=eps, infill=False)
            for k in res:
                res[k].append(o[k])
        data = dict(batch)
        data['infer_out_local_traj_tp'] = torch.stack(res['local_traj'], dim=1).permute(2, 0, 1, 3).contiguous()
        data['infer_out_trans'] = torch.stack(res['trans'], dim=1)
        data['infer_out_orient'] = torch.stack(res['orient'], dim=1)
        data['infer_out_pose'] = torch.cat([data['infer_out_orient'], pose.unsqueeze(1).expand(-1, sample_num, -1, -1)], dim=-1)
        data['batch_size'], data['seq_len'] = B, T
        return data


def _best_checkpoint(cfg_dir, version=None, cp='best'):
    """lib/utils/tools.py:41-45,94-104"""
    if version is None:
        vs = sorted(int(os.path.basename(x)[len('version_'):]) for x in glob.glob('%s/version_*' % cfg_dir))
        if not vs:
            raise FileNotFoundError('no version_* directory under %s' % cfg_dir)
        version = vs[-1]
    d = '%s/version_%s/checkpoints' % (cfg_dir, version)
    if cp == 'last':
        return '%s/last.ckpt' % d
    if cp == 'best':
        return sorted(glob.glob('%s/*best*.ckpt' % d))[-1]
    return '%s/model-epoch=%04d.ckpt' % (d, int(cp))


class MotionTrajJointModel:
    """Loads both checkpoints from `<results_root>/motion_filler/<mfiller_cfg>/version_N/checkpoints/` and
    `<results_root>/traj_pred/<trajpred_cfg>/...` (the reference's layout) and chains infiller -> trajectory predictor in ONE
    device call (motion_traj_joint_model.py:141-145)."""

    def __init__(self, cfg=None, device=torch.device('cuda'), log=None, smpl=None, results_root='results'):
        self.cfg, self.device, self.log = cfg, torch.device(device), log
        specs = getattr(cfg, 'model_specs', None) or (cfg or {}).get('model_specs', {}) if cfg is not None else {}
        specs = specs or {'mfiller_cfg': 'motion_infiller_demo', 'trajpred_cfg': 'traj_pred_demo'}
        self.specs = specs
        self.multi_step_mfiller, self.multi_step_trajpred = True, False
        if smpl is None:
            from ..lib.models.smpl import SMPL, SMPL_MODEL_DIR
            smpl = SMPL(SMPL_MODEL_DIR, pose_type='body26fk', create_transl=False)
        self.smpl = smpl
        self.mfiller_cp = _best_checkpoint(os.path.join(results_root, 'motion_filler', specs['mfiller_cfg']), specs.get('mfiller_version'), specs.get('mfiller_cp', 'best'))
        self.trajpred_cp = _best_checkpoint(os.path.join(results_root, 'traj_pred', specs['trajpred_cfg']), specs.get('trajpred_version'), specs.get('trajpred_cp', 'best'))
        if log is not None:
            log.info('loading motion infiller from check point %s' % self.mfiller_cp)
            log.info('loading trajectory predictor from check point %s' % self.trajpred_cp)
        self.mfiller = MotionInfillerVAE.load_from_checkpoint(self.mfiller_cp, cfg=None, strict=False).to(self.device).eval()
        self.traj_predictor = TrajPredVAE.load_from_checkpoint(self.trajpred_cp, cfg=None, strict=False).to(self.device).eval()
        self.traj_predictor.rest_joints = smpl.rest_joints()
        self.handle = MotionPriorsHandle(self.mfiller._sd, self.traj_predictor._sd, smpl.rest_joints(), SMPL_PARENTS, self.device)

    def get_motion_latent(self, seq_len):
        return self.mfiller.get_latent(seq_len)

    def get_traj_latent(self, seq_len):
        return self.traj_predictor.get_latent(seq_len)

    def infer_padded(self, body_pose, visible, lens, motion_eps, traj_eps):
        """Batched entry used by GlobalReconOptimizer: ragged sequences padded to a common length."""
        return self.handle.infer(body_pose, visible, lens, motion_eps=motion_eps, traj_eps=traj_eps)

    def inference(self, batch, sample_num=5, recon=False):
        if recon:
            raise NotImplementedError('recon=True needs the posterior encoders (training path)')
        pose = batch['in_body_pose'].to(self.device).float()
        vis = (batch['frame_mask'].to(self.device) == 1).float()
        B, T = pose.shape[:2]
        nw = num_windows(T)
        res = {k: [] for k in ('pose', 'local_traj', 'trans', 'orient')}
        for _ in range(sample_num):
            me = batch['in_motion_latent'].to(self.device).float().view(1, nw, NZ).expand(B, -1, -1).contiguous() if 'in_motion_latent' in batch \
                else torch.randn((B, nw, NZ), device=self.device)
            te = batch['in_traj_latent'].to(self.device).float().expand(B, -1).contiguous() if 'in_traj_latent' in batch \
                else torch.randn((B, NZ), device=self.device)
            o = self.handle.infer(pose, vis, [T] * B, motion_eps=me, traj_eps=te)
            for k in res:
                res[k].append(o[k])
        data = dict(batch)
        body = torch.stack(res['pose'], dim=1)
        data['infer_out_body_pose'] = body
        data['infer_out_orient'] = torch.stack(res['orient'], dim=1)
        data['infer_out_trans'] = torch.stack(res['trans'], dim=1)
        data['infer_out_pose'] = torch.cat([data['infer_out_orient'], body], dim=-1)
        data['infer_out_local_traj_tp'] = torch.stack(res['local_traj'], dim=1).permute(2, 0, 1, 3).contiguous()
        return data
