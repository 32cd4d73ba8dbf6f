"""SMPL body model on MI355X -- drop-in for the reference class at lib/models/smpl.py:274-343.

Same constructor and call signature:

    smpl = SMPL(SMPL_MODEL_DIR, pose_type='body26fk', create_transl=False).to(device)
    out = smpl(global_orient=(B,3), body_pose=(B,69), betas=(B,10), root_trans=(B,3), root_scale=(B,) or None,
               return_full_pose=True, orig_joints=False)      # -> ModelOutput(vertices, joints, full_pose, betas, ...)
    j = smpl.get_joints(global_orient=, body_pose=, betas=, root_trans=)          # forward kinematics only, (B,24,3)

All arithmetic (Rodrigues, kinematic chain, blend shapes, skinning, joint regression, re-anchoring) runs in hand-written
HIP kernels behind the C ABI (`glamr_smpl_*` in include/glamr_hip.h); this file only loads the model file, owns the device
handle and gives the call autograd semantics.  Gradients flow to every input: when only `global_orient`, `root_trans` and `root_scale`
need them (what the shipped configurations differentiate, global_recon_model.py:517-524,591-633) the backward uses the rigid identity on
the forward outputs (glamr_smpl_backward_root); when `body_pose` or `betas` require gradients (the latent-optimisation mode, :434-437) the
general backward runs (glamr_smpl_backward: skinning, blend shapes, kinematic chain, joint regression, re-anchoring in reverse), with or
without `root_trans`.  There is no CPU implementation: tensors must live on a HIP device.
"""
import os
import pickle
from collections import namedtuple

import numpy as np
import torch
from torch import nn

from ... import _lib

ModelOutput = namedtuple('ModelOutput', ['vertices', 'joints', 'full_pose', 'betas', 'global_orient', 'body_pose', 'expression',
                                         'left_hand_pose', 'right_hand_pose', 'jaw_pose', 'global_trans', 'scale'])
ModelOutput.__new__.__defaults__ = (None,) * len(ModelOutput._fields)

# asset locations, relative to the working directory like the reference (lib/models/smpl.py:28-31)
JOINT_REGRESSOR_TRAIN_EXTRA = 'data/J_regressor_extra.npy'
SMPL_MODEL_DIR = 'data/body_models/smpl'

# Vertices that smplx appends to the 24 kinematic joints for the SMPL topology (published `vertex_ids.py` table 'smplh'):
# nose, reye, leye, rear, lear, LBigToe, LSmallToe, LHeel, RBigToe, RSmallToe, RHeel, then left/right finger tips.
SMPLX_EXTRA_VERTEX_IDS = [332, 6260, 2800, 4071, 583, 3216, 3226, 3387, 6617, 6624, 6787,
                          2746, 2319, 2445, 2556, 2673, 6191, 5782, 5905, 6016, 6133]

# Joint name -> index into [24 chain joints | 21 picked vertices | 9 extra-regressed joints]  (lib/models/smpl.py:35-57)
_CHAIN = {'OP MidHip': 0, 'OP LHip': 1, 'OP RHip': 2, 'OP LKnee': 4, 'OP RKnee': 5, 'OP LAnkle': 7, 'OP RAnkle': 8, 'OP Neck': 12,
          'OP LShoulder': 16, 'OP RShoulder': 17, 'OP LElbow': 18, 'OP RElbow': 19, 'OP LWrist': 20, 'OP RWrist': 21}
_PICKED = ['OP Nose', 'OP REye', 'OP LEye', 'OP REar', 'OP LEar', 'OP LBigToe', 'OP LSmallToe', 'OP LHeel', 'OP RBigToe',
           'OP RSmallToe', 'OP RHeel', 'Left Thumb Tip', 'Left Index Tip', 'Left Middle Tip', 'Left Ring Tip', 'Left Pinky Tip',
           'Right Thumb Tip', 'Right Index Tip', 'Right Middle Tip', 'Right Ring Tip', 'Right Pinky Tip']
_EXTRA = ['Right Hip', 'Left Hip', 'Neck (LSP)', 'Top of Head (LSP)', 'Pelvis (MPII)', 'Thorax (MPII)', 'Spine (H36M)',
          'Jaw (H36M)', 'Head (H36M)']
JOINT_MAP = dict(_CHAIN)
JOINT_MAP.update({n: 24 + i for i, n in enumerate(_PICKED)})
JOINT_MAP.update({n: 45 + i for i, n in enumerate(_EXTRA)})

_POSE_TYPES = {
    # lib/models/smpl.py:221-250
    'body26fk': ['Pelvis (MPII)', 'OP LHip', 'OP RHip', 'Spine (H36M)', 'OP LKnee', 'OP RKnee', 'OP Neck', 'OP LAnkle', 'OP RAnkle',
                 'OP LBigToe', 'OP RBigToe', 'OP LSmallToe', 'OP RSmallToe', 'OP LHeel', 'OP RHeel', 'OP Nose', 'OP LEye', 'OP REye',
                 'OP LEar', 'OP REar', 'OP LShoulder', 'OP RShoulder', 'OP LElbow', 'OP RElbow', 'OP LWrist', 'OP RWrist'],
}


def get_ordered_joint_names(pose_type):
    return list(_POSE_TYPES[pose_type])


def _dense(x):
    return np.asarray(x.todense() if hasattr(x, 'todense') else x)


class _SmplFn(torch.autograd.Function):

    @staticmethod
    def forward(ctx, module, pose, betas, root_trans, root_scale, want_verts, orig_joints, general=False):
        h = module._handle(pose.device)
        B = pose.shape[0]
        n_out = 24 if orig_joints else module.n_out
        joints = torch.empty((B, n_out, 3), device=pose.device, dtype=torch.float32)
        verts = torch.empty((B, module.num_verts, 3), device=pose.device, dtype=torch.float32) if want_verts else None
        L = _lib.lib()
        ws = torch.empty(L.glamr_smpl_workspace_bytes(h, B), device=pose.device, dtype=torch.uint8)
        flags = 1 if orig_joints else 0
        _lib.check(L.glamr_smpl_forward(h, B, _lib.ptr(pose), _lib.ptr(betas), _lib.ptr(root_trans), _lib.ptr(root_scale),
                                        _lib.ptr(verts), _lib.ptr(joints), flags, _lib.ptr(ws), _lib.current_stream()))
        ctx.module, ctx.flags, ctx.want_verts, ctx.general = module, flags, want_verts, general
        ctx.save_for_backward(pose, root_trans, root_scale, verts, joints, betas)
        if want_verts:
            return verts, joints
        return joints.new_empty(0), joints

    @staticmethod
    def backward(ctx, g_verts, g_joints):
        pose, root_trans, root_scale, verts, joints, betas = ctx.saved_tensors
        module = ctx.module
        if ctx.general:
            return _SmplFn._backward_general(ctx, g_verts, g_joints)
        if root_trans is None:
            raise NotImplementedError('SMPL backward is implemented for the re-anchored call (root_trans given), the form the '
                                      'global optimiser uses (global_recon_model.py:517-524)')
        B = pose.shape[0]
        g_orient = torch.empty((B, 3), device=pose.device, dtype=torch.float32)
        g_trans = torch.empty((B, 3), device=pose.device, dtype=torch.float32)
        g_scale = torch.empty((B,), device=pose.device, dtype=torch.float32) if root_scale is not None else None
        gv = g_verts.contiguous() if (ctx.want_verts and g_verts is not None) else None
        gj = g_joints.contiguous() if g_joints is not None else None
        L = _lib.lib()
        _lib.check(L.glamr_smpl_backward_root(module._handle(pose.device), B, _lib.ptr(pose), _lib.ptr(root_trans), _lib.ptr(root_scale),
                                              _lib.ptr(verts), _lib.ptr(joints), _lib.ptr(gv), _lib.ptr(gj), _lib.ptr(g_orient),
                                              _lib.ptr(g_trans), _lib.ptr(g_scale), ctx.flags, _lib.current_stream()))
        g_pose = torch.zeros_like(pose)
        g_pose[:, :3] = g_orient
        return None, g_pose, None, g_trans, g_scale, None, None, None

    @staticmethod
    def _backward_general(ctx, g_verts, g_joints):
        pose, root_trans, root_scale, verts, joints, betas = ctx.saved_tensors
        module, B, dev = ctx.module, pose.shape[0], pose.device
        gv = g_verts.contiguous() if (ctx.want_verts and g_verts is not None) else None
        gj = g_joints.contiguous() if g_joints is not None else None
        if gv is None and gj is None:
            return (None,) * 8
        g_pose = torch.empty((B, 72), device=dev, dtype=torch.float32)
        g_betas = torch.empty_like(betas)
        g_trans = torch.empty((B, 3), device=dev, dtype=torch.float32) if root_trans is not None else None
        g_scale = torch.empty((B,), device=dev, dtype=torch.float32) if root_scale is not None else None
        L = _lib.lib()
        h = module._handle(dev)
        ws = torch.empty(L.glamr_smpl_backward_workspace_bytes(h, B, 1 if gv is not None else 0), device=dev, dtype=torch.uint8)
        _lib.check(L.glamr_smpl_backward(h, B, _lib.ptr(pose), _lib.ptr(betas), _lib.ptr(root_trans), _lib.ptr(root_scale), _lib.ptr(verts), _lib.ptr(joints),
                                         _lib.ptr(gv), _lib.ptr(gj), _lib.ptr(g_pose), _lib.ptr(g_betas), _lib.ptr(g_trans), _lib.ptr(g_scale), ctx.flags,
                                         _lib.ptr(ws), _lib.current_stream()))
        return None, g_pose, g_betas, g_trans, g_scale, None, None, None


class SMPL(nn.Module):
    """See the module docstring.  `model_dir` holds SMPL_{NEUTRAL,MALE,FEMALE}.pkl with the public SMPL keys
    (v_template, shapedirs, posedirs, J_regressor, weights, kintree_table, f); `data/J_regressor_extra.npy` (9 x V) is read
    relative to the working directory unless `extra_regressor_path` is given."""

    NUM_JOINTS = 23
    NUM_BODY_JOINTS = 23

    def __init__(self, model_path=SMPL_MODEL_DIR, *args, pose_type=None, gender='neutral', num_betas=10, create_transl=False,
                 extra_regressor_path=None, joint_names=None, **kwargs):
        super().__init__()
        path = model_path
        if os.path.isdir(path):
            cand = os.path.join(path, 'SMPL_%s.pkl' % gender.upper())
            path = cand if os.path.exists(cand) else os.path.join(path, 'SMPL_NEUTRAL.pkl')
        with open(path, 'rb') as f:
            md = pickle.load(f, encoding='latin1')
        f32 = lambda x: np.ascontiguousarray(_dense(x), dtype=np.float32)
        self.num_betas = num_betas
        self._v_template = f32(md['v_template'])
        self._shapedirs = np.ascontiguousarray(f32(md['shapedirs'])[:, :, :num_betas])
        pd = f32(md['posedirs'])
        self._posedirs = np.ascontiguousarray(pd.reshape(-1, pd.shape[-1]).T)                # (207, V*3), smplx layout
        self._J_regressor = f32(md['J_regressor'])
        self._lbs_weights = f32(md['weights'])
        parents = _dense(md['kintree_table'])[0].astype(np.int64)
        parents[0] = -1
        self.parents = torch.tensor(parents, dtype=torch.long)
        self.faces = _dense(md['f']).astype(np.int64)
        self.num_verts = self._v_template.shape[0]
        self._J_extra = np.ascontiguousarray(np.load(extra_regressor_path or JOINT_REGRESSOR_TRAIN_EXTRA), dtype=np.float32)
        if joint_names is None:
            joint_names = get_ordered_joint_names(pose_type if pose_type is not None else 'body26fk')
        self.joint_names = joint_names
        self.joint_map = torch.tensor([JOINT_MAP[n] for n in joint_names], dtype=torch.long)
        self.n_out = len(joint_names)
        # host-visible copies other GLAMR code reads (evaluator, visualiser, trajectory predictor FK)
        self.register_buffer('J_regressor_extra', torch.from_numpy(self._J_extra), persistent=False)
        self.register_buffer('v_template', torch.from_numpy(self._v_template), persistent=False)
        self.register_buffer('J_regressor', torch.from_numpy(self._J_regressor), persistent=False)
        self.register_buffer('faces_tensor', torch.tensor(self.faces, dtype=torch.long), persistent=False)
        self._handles = {}

    # -- device handle ---------------------------------------------------------------------------------------------------
    def _handle(self, device):
        if device.type != 'cuda':
            raise RuntimeError('glamr_amd SMPL runs on an MI355X only (got a %s tensor); there is no CPU fallback' % device.type)
        key = device.index if device.index is not None else torch.cuda.current_device()
        if key not in self._handles:
            import ctypes
            L = _lib.lib()
            h = ctypes.c_void_p()
            with torch.cuda.device(key):
                _lib.check(L.glamr_smpl_create(
                    ctypes.byref(h), self.num_verts, self.num_betas, _lib.ptr(self._v_template), _lib.ptr(self._shapedirs),
                    _lib.ptr(self._posedirs), _lib.ptr(self._J_regressor), _lib.ptr(self._lbs_weights), _lib.ptr(self._J_extra),
                    self._J_extra.shape[0], _lib.ptr(self.parents.numpy().astype(np.int32)),
                    _lib.ptr(np.asarray(SMPLX_EXTRA_VERTEX_IDS, dtype=np.int32)), len(SMPLX_EXTRA_VERTEX_IDS),
                    _lib.ptr(self.joint_map.numpy().astype(np.int32)), self.n_out))
            self._handles[key] = h
        return self._handles[key]

    def __del__(self):
        try:
            for h in self._handles.values():
                _lib.lib().glamr_smpl_destroy(h)
        except Exception:
            pass

    def rest_joints(self):
        """(24,3) J_regressor @ v_template, the unshaped rest skeleton used by get_joints (smpl.py:327)."""
        return (self._J_regressor.astype(np.float64) @ self._v_template.astype(np.float64)).astype(np.float32)

    # -- reference API ---------------------------------------------------------------------------------------------------
    def forward(self, *args, betas=None, body_pose=None, global_orient=None, root_trans=None, root_scale=None, orig_joints=False,
                return_full_pose=False, return_verts=True, get_skin=True, **kwargs):
        if args:
            # smplx.SMPL.forward(betas, body_pose, global_orient, transl, ...): the positional order of the class the reference extends
            names = ('betas', 'body_pose', 'global_orient')
            if len(args) > len(names):
                raise TypeError('SMPL.forward takes at most betas, body_pose, global_orient positionally (transl is not used by GLAMR)')
            given = dict(zip(names, args))
            betas = given.get('betas', betas)
            body_pose = given.get('body_pose', body_pose)
            global_orient = given.get('global_orient', global_orient)
        general = bool((body_pose.requires_grad or betas.requires_grad or (root_trans is None and global_orient.requires_grad)) and torch.is_grad_enabled())
        pose = torch.cat([global_orient, body_pose], dim=1).float().contiguous()
        B = pose.shape[0]
        if betas.shape[0] != B:
            betas = betas.expand(B, -1)
        betas = betas.float().contiguous()
        rt = root_trans.float().contiguous() if root_trans is not None else None
        rs = root_scale.float().contiguous() if root_scale is not None else None
        verts, joints = _SmplFn.apply(self, pose, betas, rt, rs, bool(return_verts), bool(orig_joints), general)
        return ModelOutput(vertices=verts if return_verts else None, joints=joints, betas=betas, global_orient=global_orient,
                           body_pose=body_pose, full_pose=pose if return_full_pose else None)

    def root_relative_joints(self, body_pose, betas):
        """forward(global_orient=0, body_pose, betas, root_trans=0, return_verts=False).joints -- the joints the optimiser caches for the whole
        schedule -- without the (B,72) concatenation and the two zero arrays per call: GLAMR_SMPL_BODY_POSE_ONLY, the
        origin from a zero array kept by this module.  body_pose (B,69), betas (B,10) float32 contiguous on the device.  No autograd."""
        B, dev = body_pose.shape[0], body_pose.device
        if body_pose.dtype != torch.float32 or betas.dtype != torch.float32 or not body_pose.is_contiguous() or not betas.is_contiguous() \
                or tuple(body_pose.shape) != (B, 69) or betas.shape[0] != B:
            raise ValueError('root_relative_joints takes contiguous float32 body_pose (B,69) and betas (B,num_betas)')
        z = self.__dict__.get('_origin')
        if z is None or z.shape[0] < B or z.device != dev:
            if torch.cuda.is_current_stream_capturing() and z is not None:
                raise RuntimeError('root_relative_joints: the origin array must not grow inside a stream capture')
            z = self.__dict__['_origin'] = torch.zeros((B, 3), dtype=torch.float32, device=dev)
            torch.cuda.current_stream(dev).synchronize()          # (once: other streams read it without an event of this fill)
        h, L = self._handle(dev), _lib.lib()
        joints = torch.empty((B, self.n_out, 3), device=dev, dtype=torch.float32)
        ws = torch.empty(L.glamr_smpl_workspace_bytes(h, B), device=dev, dtype=torch.uint8)
        _lib.check(L.glamr_smpl_forward(h, B, _lib.ptr(body_pose), _lib.ptr(betas), _lib.ptr(z), None, None, _lib.ptr(joints), 2, _lib.ptr(ws),
                                        _lib.current_stream()))
        return joints

    def get_joints(self, betas=None, body_pose=None, global_orient=None, transl=None, pose2rot=True, root_trans=None, root_scale=None,
                   dtype=torch.float32):
        """Forward kinematics of the 24 chain joints from the unshaped template (betas are ignored, as in the reference)."""
        assert pose2rot, 'rotation-matrix input is not used on the GLAMR hot path'
        pose = torch.cat([global_orient, body_pose], dim=1).float().contiguous()
        B = pose.shape[0]
        joints = torch.empty((B, 24, 3), device=pose.device, dtype=torch.float32)
        rt = root_trans.float().contiguous() if root_trans is not None else None
        rs = root_scale.float().contiguous() if root_scale is not None else None
        _lib.check(_lib.lib().glamr_smpl_fk(self._handle(pose.device), B, _lib.ptr(pose), _lib.ptr(rt), _lib.ptr(rs), _lib.ptr(joints),
                                            _lib.current_stream()))
        if transl is not None and root_trans is None:      # a translation applied before re-anchoring cancels out (smpl.py:334-341)
            joints = joints + transl.unsqueeze(1)
        return joints
