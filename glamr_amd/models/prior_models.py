"""Drop-in host classes of the two motion priors and their wrapper.  Same names, constructor arguments, `load_from_checkpoint`,
`.inference(...)` signatures and output dictionary keys as the reference:

    MotionInfillerVAE      motion_infiller/models/motion_infiller_vae.py:440-667
    TrajPredVAE            traj_pred/models/traj_pred_vae.py:341-548
    MotionTrajJointModel   motion_infiller/models/motion_traj_joint_model.py:17-145

The networks themselves run in HIP kernels (glamr_nets_* ABI); these classes hold the checkpoint tensors, draw the Gaussian
latents when the caller does not supply them (lib/utils/dist.py:21-23) and arrange outputs in the reference's layouts (the mutable
`data` dictionary with its `*_tp` time-major keys).  `forward(data)` (context encoder + posterior encoder + decoder in
'train' mode) and `inference(recon=True)` run on the device as well (glamr_nets_infiller_window / glamr_nets_traj_clip); the Lightning
training hooks (`training_step`, optimisers, losses) are out of scope.
"""
import glob
import os

import numpy as np
import torch

from .layouts import INFILLER_LAYOUT, TRAJPRED_LAYOUT
from .priors import MotionPriorsHandle, num_windows, local_to_global, NZ, PAST, CUR, VAE_INFER, VAE_TRAIN, VAE_RECON
from ..lib.utils.dist import Normal

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

    noise_device = None      # where torch.randn draws the latents' noise (None = the model device, as torch.randn_like does in the reference)

    def _randn(self, shape):
        dev = self.device if self.noise_device is None else torch.device(self.noise_device)
        return torch.randn(shape, device=dev).to(self.device)

    def __call__(self, data):
        return self.forward(data)


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

    def init_batch_data(self, batch):
        """motion_infiller_vae.py:495-549 (use_joints False, axis-angle, no pose dropout at inference): layout only."""
        data = dict(batch)
        to = lambda t: t.to(self.device) if torch.is_tensor(t) else t
        for k in list(data.keys()):
            data[k] = to(data[k])
        data['invis_frame_mask'] = data['frame_mask'] == 1
        data['vis_frame_mask'] = ~data['invis_frame_mask']              # True = frame NOT visible (key-padding mask)
        if 'frame_loss_mask' in data:
            data['frame_loss_mask_tp'] = data['frame_loss_mask'].transpose(0, 1)
        if 'pose' in data:
            data['pose_tp'] = data['pose'].transpose(0, 1).contiguous()
            data['body_pose_tp'] = data['pose_tp'][..., 3:]
        if 'pose_mask' in data:
            data['pose_mask_tp'] = data['pose_mask'].transpose(0, 1).contiguous()
        if 'in_pose' not in data:
            if 'pose' in data:
                data['in_pose_tp'] = data['pose_tp'] * data['pose_mask_tp']
        else:
            data['in_pose_tp'] = data['in_pose'].transpose(0, 1).contiguous()
        if 'in_body_pose' not in data:
            data['in_body_pose_tp'] = data['in_pose_tp'][..., 3:]
        else:
            data['in_body_pose_tp'] = data['in_body_pose'].transpose(0, 1).contiguous()
        data['batch_size'] = data['in_body_pose_tp'].shape[1]
        data['seq_len'] = data['in_body_pose_tp'].shape[0]
        return data

    def _window_pass(self, data, mode, eps=None, _handle=None):
        """context encoder (+ posterior) + prior + decoder on ONE window held in `data` (time-major keys).  Fills the keys the reference's
        three sub-modules write (:107-123, :204-249, :345-433) for this mode."""
        h = _handle or self._ensure_handle()
        T, B = data['in_body_pose_tp'].shape[:2]
        win = self.past_nframe + self.cur_nframe + self.fut_nframe
        if T != win:
            raise ValueError('the motion infiller works on windows of %d frames (got %d); use inference(..., multi_step=True) for sequences' % (win, T))
        fm = (~data['vis_frame_mask']).float()
        body = data['body_pose_tp'].transpose(0, 1) if mode != VAE_INFER else None
        o = h.infiller_window(mode, data['in_body_pose_tp'].transpose(0, 1), fm, eps=eps, body_pose=body)
        data['x_in'] = data['in_body_pose_tp']
        data['context'] = o['context'].transpose(0, 1).contiguous()
        name = {VAE_INFER: 'infer', VAE_TRAIN: 'train', VAE_RECON: 'recon'}[mode]
        data['p_z_dist' + ('_infer' if mode == VAE_INFER else '')] = Normal(mu=o['p_z'][:, 0], logvar=o['p_z'][:, 1])
        if mode != VAE_INFER:
            data['q_z_dist'] = Normal(mu=o['q_z'][:, 0], logvar=o['q_z'][:, 1])
            if mode == VAE_TRAIN:
                data['q_z_samp'] = o['z']
        x = torch.cat([data['x_in'][:self.past_nframe], o['out_body_pose'].transpose(0, 1)], dim=0)       # :398 (T - fut, B, 69)
        root = data['pose_tp'][:-self.fut_nframe, :, :3] if 'pose_tp' in data else torch.zeros_like(x[..., :3])
        if mode == VAE_INFER:
            x, root = x.unsqueeze(2), root.unsqueeze(2)
        data[name + '_out_body_pose_tp'] = x
        data[name + '_out_pose_tp'] = torch.cat((root, x), dim=-1)                                        # :415-418
        return data

    def forward(self, data):
        """The training-mode pass (:478-482): data = init_batch_data(batch) of 50-frame windows.  Adds context, q_z_dist, q_z_samp,
        p_z_dist, train_out_body_pose_tp (40,B,69), train_out_pose_tp (40,B,72)."""
        B = data['in_body_pose_tp'].shape[1]
        return self._window_pass(data, VAE_TRAIN, eps=self._randn((B, self.nz)))

    def _multi_step_recon(self, batch, _handle=None):
        """inference_multi_step(recon=True) :618-632: sliding windows, the posterior MODE decoded, each window's output written back into
        the running input before the next window is cut."""
        data = self.init_batch_data(batch)
        P, C, Fu = self.past_nframe, self.cur_nframe, self.fut_nframe
        total = data['seq_len']
        tp_keys = [k for k in data if 'tp' in k]
        for i in range(int(np.ceil((total - P) / C))):
            s, e = i * C, i * C + P + C + Fu
            eb = min(e, total)
            w = {'batch_size': data['batch_size'], 'seq_len': e - s}
            for k in tp_keys:                                                                             # get_seg_data :564-587
                v = data[k][s:eb].clone()
                if e > eb:
                    v = torch.cat([v, torch.zeros((e - eb,) + v.shape[1:], device=v.device, dtype=v.dtype)], dim=0)
                w[k] = v
            m = data['vis_frame_mask'][:, s:eb].clone()
            if e > eb:
                m = torch.cat([m, torch.ones(m.shape[:-1] + (e - eb,), device=m.device, dtype=m.dtype)], dim=1)
            m[:, :P] = False
            w['vis_frame_mask'] = m
            self._window_pass(w, VAE_RECON, _handle=_handle)
            w['recon_out_pose'] = w['recon_out_pose_tp'].transpose(1, 0).contiguous()
            w['recon_out_body_pose'] = w['recon_out_pose'][..., 3:]
            nf = min(e - Fu, total) - s
            for key in ('pose', 'body_pose'):                                                             # get_res_from_cur_data :589-602
                if 'in_%s_tp' % key in data:
                    data['in_%s_tp' % key][s:s + nf] = w['recon_out_%s_tp' % key][:nf]
                    if 'recon_out_' + key not in data:
                        data['recon_out_' + key] = w['recon_out_' + key][:, :nf]
                    else:
                        data['recon_out_' + key] = torch.cat([data['recon_out_' + key], w['recon_out_' + key][:, P:nf]], dim=1)
        return data

    def inference(self, batch, sample_num=5, recon=False, multi_step=False, _handle=None):
        """:643-667.  batch: {'in_body_pose' (B,T,69) or 'pose' + 'pose_mask', 'frame_mask' (B,T) 1 = visible[, 'in_motion_latent']}.
        multi_step=True (the call GLAMR makes): sliding windows over a sequence -> infer_out_body_pose (B,S,T,69), infer_out_pose (B,S,T,72);
        multi_step=False: the batch IS one 50-frame window -> (B,S,40,.).  recon=True adds recon_out_pose / recon_out_body_pose."""
        h = _handle or self._ensure_handle()
        if not multi_step:
            data = self.init_batch_data(batch)
            B = data['batch_size']
            rep = lambda t, dim: t.repeat_interleave(sample_num, dim=dim)
            w = dict(data)
            w['in_body_pose_tp'] = rep(data['in_body_pose_tp'], 1)
            w['vis_frame_mask'] = rep(data['vis_frame_mask'], 0)
            if 'pose_tp' in data:
                w['pose_tp'] = rep(data['pose_tp'], 1)
            eps = data['in_motion_latent'].to(self.device).float() if 'in_motion_latent' in data else self._randn((B * sample_num, self.nz))
            if eps.shape[0] != B * sample_num:
                eps = eps.expand(B * sample_num, -1)
            self._window_pass(w, VAE_INFER, eps=eps.contiguous(), _handle=h)
            data['x_in'], data['context'] = data['in_body_pose_tp'], w['context'][:, ::sample_num].contiguous()
            data['p_z_dist_infer'] = w['p_z_dist_infer']
            for k in ('infer_out_body_pose_tp', 'infer_out_pose_tp'):
                data[k] = w[k].squeeze(2).reshape(w[k].shape[0], B, sample_num, -1)
            data['infer_out_pose'] = data['infer_out_pose_tp'].permute(1, 2, 0, 3).contiguous()
            data['infer_out_body_pose'] = data['infer_out_pose'][..., 3:]
            if recon:
                self._window_pass(data, VAE_RECON, _handle=h)
                data['recon_out_pose'] = data['recon_out_pose_tp'].transpose(1, 0).contiguous()
                data['recon_out_body_pose'] = data['recon_out_pose'][..., 3:]
            for k in ('pose', 'trans', 'shape'):                                                          # :664-666
                if k in data:
                    data[k] = data[k][:, :-self.fut_nframe]
            return data
        if 'in_body_pose' in batch:
            pose = batch['in_body_pose'].to(self.device).float()
        else:
            pose = (batch['pose'] * batch['pose_mask']).to(self.device).float()[..., 3:]
        vis = (batch['frame_mask'].to(self.device) == 1).float()
        B, T = pose.shape[:2]
        nw = num_windows(T)
        outs = []
        for _ in range(sample_num):
            if 'in_motion_latent' in batch:
                eps = batch['in_motion_latent'].to(self.device).float().view(1, nw, NZ).expand(B, -1, -1).contiguous()
            else:
                eps = self._randn((B, nw, NZ))
            outs.append(h.infer(pose.contiguous(), vis, [T] * B, motion_eps=eps, traj=False)['pose'])
        body = torch.stack(outs, dim=1)                                   # (B,S,T,69)
        data = dict(batch)
        data['infer_out_body_pose'] = body
        root = batch['pose'].to(self.device).float()[..., :3].unsqueeze(1).expand(-1, sample_num, -1, -1) if 'pose' in batch else torch.zeros_like(body[..., :3])
        data['infer_out_pose'] = torch.cat([root, body], dim=-1)             # the root rides along from pose_tp when the batch has it (:413-418)
        data['in_body_pose_tp'] = outs[-1].transpose(0, 1).contiguous()   # the running input ends up holding the last sample
        data['batch_size'], data['seq_len'] = B, T
        if recon:
            r = self._multi_step_recon(batch, _handle=h)
            data['recon_out_pose'], data['recon_out_body_pose'] = r['recon_out_pose'], r['recon_out_body_pose']
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

    seq_len = 100            # chunk length of the multi-step (chunked) inference, traj_pred_demo.yml

    def get_joint_pos(self, body_pose):
        """:384-394 -- forward kinematics of the 23 body joints for zero shape / root orientation, relative to the root."""
        from .. import _lib
        h = self._ensure_handle()
        flat = body_pose.reshape(-1, 1, 69).to(self.device).float().contiguous()
        # the FK runs inside glamr_nets_traj_clip as well; this stand-alone call serves `joint_pos_tp` of init_batch_data
        from ..lib.utils import np_transform as nt  # noqa: F401  (host one-offs live there; FK itself is a device kernel)
        if self._smpl is None:
            raise RuntimeError('get_joint_pos needs the SMPL body model: construct through MotionTrajJointModel or set .smpl')
        z3 = torch.zeros((flat.shape[0], 3), device=self.device)
        j = self._smpl.get_joints(global_orient=z3, body_pose=flat[:, 0], betas=torch.zeros((flat.shape[0], 10), device=self.device), root_trans=z3)
        return j[:, 1:, :].reshape(body_pose.shape[:-1] + (-1,))

    _smpl = None

    def init_batch_data(self, batch):
        """:396-457 -- layout, plus the derived tensors the sub-modules read: `local_traj_tp` (traj_global2local_heading) comes from the
        device pass (filled by forward / inference), `orient_q_tp` from the host one-off converter."""
        from ..lib.utils import np_transform as nt
        data = dict(batch)
        for k in list(data.keys()):
            if torch.is_tensor(data[k]):
                data[k] = data[k].to(self.device)
        if 'pose' in data:
            data['pose_tp'] = data['pose'].transpose(0, 1).contiguous()
            data['body_pose_tp'] = data['pose_tp'][..., 3:]
            data['orient_tp'] = data['pose_tp'][..., :3]
        if 'in_pose' in data:
            data['in_pose_tp'] = data['in_pose'].transpose(0, 1).contiguous()
        elif 'pose' in data:
            data['in_pose_tp'] = data['pose_tp']
        if 'in_body_pose' in data:
            data['in_body_pose_tp'] = data['in_body_pose'].transpose(0, 1).contiguous()
        elif 'in_pose_tp' in data:
            data['in_body_pose_tp'] = data['in_pose_tp'][..., 3:]
        if 'trans' in data:
            data['trans_tp'] = data['trans'].transpose(0, 1).contiguous()
            data['orient_q_tp'] = torch.from_numpy(nt.aa_to_quat(data['orient_tp'].cpu().numpy())).to(self.device)
        if 'in_joint_pos' in data:
            data['in_joint_pos_tp'] = data['in_joint_pos'].transpose(0, 1).contiguous()
        ref = data.get('in_joint_pos_tp', data.get('in_body_pose_tp'))
        data['batch_size'], data['seq_len'] = ref.shape[1], ref.shape[0]
        return data

    def _clip_pass(self, data, mode, eps=None, sample_num=1, valid_len=0, _handle=None):
        """context encoder (+ posterior) + prior + decoder on one clip held in `data`; fills the keys of :72-92, :160-199, :269-334."""
        h = _handle or self._ensure_handle()
        bm = lambda k: data[k].transpose(0, 1).contiguous() if k in data else None
        S = sample_num if mode == VAE_INFER else 1
        rep = lambda t: None if t is None else (t.repeat_interleave(S, dim=0) if S > 1 else t)
        kw = dict(in_joint_pos=rep(bm('in_joint_pos_tp'))) if 'in_joint_pos_tp' in data else dict(in_body_pose=rep(bm('in_body_pose_tp')))
        init_row = None
        if 'init_xy' in data:                                                                             # DataDecoder :319-321
            from ..lib.utils import np_transform as nt
            init_row = torch.zeros((data['init_xy'].shape[0], 11), device=self.device)
            init_row[:, :2] = data['init_xy']
            init_row[:, 9:] = torch.from_numpy(nt.heading_to_vec(data['init_heading'].cpu().numpy()).astype(np.float32)).to(self.device)
            init_row = rep(init_row)
        o = h.traj_clip(mode, trans=rep(bm('trans_tp')), orient=rep(bm('orient_tp')) if 'trans_tp' in data else None, eps=eps, valid_len=valid_len,
                        init_row=init_row, **kw)
        name = {VAE_INFER: 'infer', VAE_TRAIN: 'train', VAE_RECON: 'recon'}[mode]
        B = data['batch_size']
        if 'local_traj' in o:
            data['local_traj_tp'] = o['local_traj'][::S].transpose(0, 1).contiguous()
        data['p_z_dist' + ('_infer' if mode == VAE_INFER else '')] = Normal(params=o['p_z'])
        if mode != VAE_INFER:
            data['q_z_dist'] = Normal(params=o['q_z'])
            if mode == VAE_TRAIN:
                data['q_z_samp'] = o['z']
        def tm(t):                                        # (B*S, T, C) -> (T, B, S, C) or (T, B, C)
            t = t.transpose(0, 1)
            return t.reshape(t.shape[0], B, S, t.shape[-1]) if mode == VAE_INFER else t.contiguous()
        data[name + '_orig_out_local_traj_tp'] = tm(o['out_orig_local_traj'])
        data[name + '_out_local_traj_tp'] = tm(o['out_local_traj'])
        data[name + '_out_trans_tp'] = tm(o['out_trans'])
        data[name + '_out_orient_q_tp'] = tm(o['out_orient_q'])
        data['_' + name + '_out_orient_tp'] = tm(o['out_orient'])         # axis-angle of the same rotation (convert_out_pose_trans :459-474)
        return data

    def forward(self, data):
        """The training-mode pass (:378-382): data = init_batch_data(batch) with pose + trans.  Adds local_traj_tp, q_z_dist, q_z_samp,
        p_z_dist, train_out_local_traj_tp (T,B,11), train_out_trans_tp (T,B,3), train_out_orient_q_tp (T,B,4)."""
        return self._clip_pass(data, VAE_TRAIN, eps=self._randn((data['batch_size'], self.nz)))

    def _convert_out(self, data, mode, sample_num=1):
        """convert_out_pose_trans :459-474"""
        if mode == 'infer':
            data['infer_out_orient_tp'] = data.pop('_infer_out_orient_tp')
            data['infer_out_orient'] = data['infer_out_orient_tp'].permute(1, 2, 0, 3).contiguous()
            data['infer_out_trans'] = data['infer_out_trans_tp'].permute(1, 2, 0, 3).contiguous()
            if 'in_body_pose_tp' in data:
                data['infer_out_pose_tp'] = torch.cat([data['infer_out_orient_tp'], data['in_body_pose_tp'].unsqueeze(2).repeat(1, 1, sample_num, 1)], dim=-1)
                data['infer_out_pose'] = data['infer_out_pose_tp'].permute(1, 2, 0, 3).contiguous()
        else:
            data['recon_out_orient_tp'] = data.pop('_recon_out_orient_tp')
            data['recon_out_orient'] = data['recon_out_orient_tp'].transpose(1, 0).contiguous()
            data['recon_out_trans'] = data['recon_out_trans_tp'].transpose(1, 0).contiguous()
            if 'in_body_pose_tp' in data:
                data['recon_out_pose_tp'] = torch.cat([data['recon_out_orient_tp'], data['in_body_pose_tp']], dim=-1)
                data['recon_out_pose'] = data['recon_out_pose_tp'].transpose(1, 0).contiguous()

    def _eps_infer(self, batch, B, sample_num):
        if 'in_traj_latent' in batch:
            e = batch['in_traj_latent'].to(self.device).float()
            return (e.expand(B * sample_num, -1) if e.shape[0] != B * sample_num else e).contiguous()
        return self._randn((B * sample_num, self.nz))

    def _multi_step(self, batch, sample_num, recon, _handle=None):
        """inference_multi_step :508-519: independent chunks of `seq_len` frames (the last one zero-padded), local rows concatenated with
        the heading carried over (:498-506), then ONE local -> global pass."""
        from ..lib.utils import np_transform as nt
        mode, name = (VAE_RECON, 'recon') if recon else (VAE_INFER, 'infer')
        data = self.init_batch_data(batch)
        if 'in_joint_pos_tp' not in data:
            data['in_joint_pos_tp'] = self.get_joint_pos(data['in_body_pose_tp'])
        total, L = data['seq_len'], self.seq_len
        tp_keys = [k for k in data if 'tp' in k and 'out' not in k]
        rows = None
        for i in range(int(np.ceil(total / L))):
            s, e = i * L, (i + 1) * L
            eb = min(e, total)
            c = {'batch_size': data['batch_size'], 'seq_len': L}
            for k in tp_keys:
                v = data[k][s:eb].clone()
                if e > eb:
                    v = torch.cat([v, torch.zeros((e - eb,) + v.shape[1:], device=v.device, dtype=v.dtype)], dim=0)
                c[k] = v
            S = 1 if recon else sample_num
            self._clip_pass(c, mode, eps=None if recon else self._eps_infer(batch, data['batch_size'], sample_num), sample_num=S, _handle=_handle)
            nf = eb - s
            if rows is None:
                rows = c[name + '_out_local_traj_tp'][:nf]
            else:
                cur = c[name + '_orig_out_local_traj_tp'].clone()
                last6 = rows[-1, ..., 3:-2].cpu().numpy()
                hv = nt.heading_to_vec(nt.heading_of(nt.rotmat_to_quat(nt.sixd_to_rotmat(last6))))
                cur[0, ..., 9:] = torch.from_numpy(hv.astype(np.float32)).to(cur.device)
                rows = torch.cat([rows, cur[:nf]], dim=0)
        data[name + '_out_local_traj_tp'] = rows
        flat = rows.reshape(rows.shape[0], -1, 11).transpose(0, 1)
        trans, orient, q = local_to_global(flat)
        back = lambda t: t.transpose(0, 1).reshape(rows.shape[:-1] + (t.shape[-1],))
        data[name + '_out_trans_tp'], data[name + '_out_orient_q_tp'], data['_' + name + '_out_orient_tp'] = back(trans), back(q), back(orient)
        return data

    def inference(self, batch, sample_num=5, recon=False, recon_only=False, multi_step=False, _handle=None):
        """:524-548.  batch: {'in_body_pose' (B,T,69) | 'pose' (B,T,72) [+ 'trans' (B,T,3) for recon][, 'in_traj_latent' (B,128)]}.
        Returns infer_out_local_traj_tp (T,B,S,11), infer_out_trans / infer_out_orient (B,S,T,3), infer_out_pose (B,S,T,72); with
        recon=True also recon_out_local_traj_tp (T,B,11), recon_out_trans / recon_out_orient (B,T,3), recon_out_pose."""
        h = _handle or self._ensure_handle()
        if multi_step:
            data = None
            if not recon_only:
                data = self._multi_step(batch, sample_num, recon=False, _handle=h)
                self._convert_out(data, 'infer', sample_num)
            if recon:
                r = self._multi_step(batch, sample_num, recon=True, _handle=h)
                if recon_only:
                    data = r
                else:
                    for k in ('recon_out_orient_q_tp', 'recon_out_trans_tp', 'recon_out_local_traj_tp', '_recon_out_orient_tp'):
                        data[k] = r[k]
                    self._convert_out(data, 'recon')
            return data
        if not recon and not recon_only and 'trans' not in batch and 'pose' not in batch and 'init_xy' not in batch and 'in_body_pose' in batch:
            # the call GLAMR makes: one batched launch per sample
            pose = batch['in_body_pose'].to(self.device).float()
            B, T = pose.shape[:2]
            res = {k: [] for k in ('local_traj', 'trans', 'orient')}
            for _ in range(sample_num):
                eps = batch['in_traj_latent'].to(self.device).float().expand(B, -1).contiguous() if 'in_traj_latent' in batch else self._randn((B, NZ))
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
        data = self.init_batch_data(batch)
        if not recon_only:
            self._clip_pass(data, VAE_INFER, eps=self._eps_infer(batch, data['batch_size'], sample_num), sample_num=sample_num, _handle=h)
            self._convert_out(data, 'infer', sample_num)
        if recon:
            self._clip_pass(data, VAE_RECON, _handle=h)
            self._convert_out(data, 'recon')
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
        self.traj_predictor._smpl = smpl
        self.handle = MotionPriorsHandle(self.mfiller._sd, self.traj_predictor._sd, smpl.rest_joints(), SMPL_PARENTS, self.device)
        self.mfiller._handle = self.traj_predictor._handle = self.handle          # one set of device weights serves all three classes

    def get_motion_latent(self, seq_len):
        return self.mfiller.get_latent(seq_len)

    def get_traj_latent(self, seq_len):
        return self.traj_predictor.get_latent(seq_len)

    def infer_padded(self, body_pose, visible, lens, motion_eps, traj_eps, buffers=None, coschedule=False, between=None):
        """Batched entry used by GlobalReconOptimizer: ragged sequences padded to a common length.  `between(out)`: called after the motion infiller
        has been enqueued (out['pose'] holds its result in stream order) and before the trajectory predictor is (two library calls instead of
        one) -- where a pipelined caller runs what only needs the infilled poses (GlobalReconOptimizer.init_resident: the skinning)."""
        if between is None:
            return self.handle.infer(body_pose, visible, lens, motion_eps=motion_eps, traj_eps=traj_eps, buffers=buffers, coschedule=coschedule)
        out = self.handle.infer(body_pose, visible, lens, motion_eps=motion_eps, traj=False, buffers=buffers, coschedule=coschedule)
        between(out)
        out.update(self.handle.infer(out['pose'], None, lens, traj_eps=traj_eps, infill=False, buffers=buffers, coschedule=coschedule))
        return out

    def pred_trajectory(self, data, sample_num, recon=False, multi_step=False):
        """motion_traj_joint_model.py:73-133 (in_joint_pos_only False, model_type 'angle'): the infiller's motion through the trajectory
        predictor, for the sampled motions and -- with recon -- for the reconstructed one."""
        from ..lib.utils import np_transform as nt
        for mode in (['infer', 'recon'] if recon else ['infer']):
            motion = data['%s_out_body_pose' % mode]
            if 'pose' in data:
                data['init_xy'] = data['trans'][:, 0, :2]
                q = nt.quat_mul(nt.aa_to_quat(data['pose'][:, 0, :3].cpu().numpy()), nt.quat_conj(np.array([0.5, 0.5, 0.5, 0.5], np.float32))[None])
                data['init_heading'] = torch.from_numpy(nt.heading_of(q).astype(np.float32)).to(self.device)
            if mode == 'infer':
                motion = motion.reshape(-1, *motion.shape[-2:])
                batch = {'in_body_pose': motion}
                if 'in_traj_latent' in data:
                    batch['in_traj_latent'] = data['in_traj_latent']
                if 'init_xy' in data:
                    batch['init_xy'] = data['init_xy'].repeat_interleave(sample_num, dim=0)
                    batch['init_heading'] = data['init_heading'].repeat_interleave(sample_num, dim=0)
                out = self.traj_predictor.inference(batch, sample_num=1, recon=False, multi_step=multi_step)
                for key in ('infer_out_pose', 'infer_out_trans', 'infer_out_orient'):
                    if key in out:
                        data[key] = out[key].reshape(-1, sample_num, *out[key].shape[-2:])
                lt = out['infer_out_local_traj_tp']
                data['infer_out_local_traj_tp'] = lt.reshape(lt.shape[0], -1, sample_num, lt.shape[-1])
            else:
                batch = {'in_body_pose': motion, 'pose': data['pose'], 'trans': data['trans']}
                out = self.traj_predictor.inference(batch, sample_num=1, recon=True, recon_only=True, multi_step=multi_step)
                for key in ('recon_out_pose', 'recon_out_trans', 'recon_out_orient', 'recon_out_local_traj_tp'):
                    if key in out:
                        data[key] = out[key]

    def inference(self, batch, sample_num=5, recon=False):
        """motion_traj_joint_model.py:141-145"""
        if recon or 'pose' in batch or 'in_body_pose' not in batch:
            data = self.mfiller.inference(batch, sample_num, recon, self.multi_step_mfiller)
            self.pred_trajectory(data, sample_num, recon, self.multi_step_trajpred)
            return data
        # the call GLAMR makes (global_recon_model.py:353-368): both networks in ONE device call per sample
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
