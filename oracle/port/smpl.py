"""TEST INFRASTRUCTURE ONLY -- CPU restatement of GLAMR's SMPL wrapper (/root/reference/lib/models/smpl.py:274-343) on top of
the smplx restatement in oracle/smplx_lbs.py."""
from collections import namedtuple
import numpy as np
import torch
from oracle.smplx_lbs import SMPLLayer, vertices2joints, batch_rodrigues, batch_rigid_transform

ModelOutput = namedtuple('ModelOutput', ['vertices', 'joints', 'full_pose', 'betas', 'global_orient', 'body_pose'])
ModelOutput.__new__.__defaults__ = (None,) * len(ModelOutput._fields)

# `body26fk` joints as indices into [24 chain joints | 21 picked vertices | 9 extra-regressed joints]
# (lib/models/smpl.py:35-57 JOINT_MAP looked up through the name list at :221-250)
BODY26FK_MAP = [49, 1, 2, 51, 4, 5, 12, 7, 8, 29, 32, 30, 33, 31, 34, 24, 26, 25, 28, 27, 16, 17, 18, 19, 20, 21]


class SMPL(SMPLLayer):
    """smpl.py:274-316.  `forward` = smplx LBS + 9 extra regressed joints + 26-joint selection + root re-anchoring."""

    def __init__(self, model_dir, pose_type='body26fk', extra_regressor_path='data/J_regressor_extra.npy', **kwargs):
        super().__init__(model_dir, **kwargs)
        assert pose_type == 'body26fk'
        self.register_buffer('J_regressor_extra', torch.tensor(np.load(extra_regressor_path), dtype=torch.float32))
        self.joint_map = torch.tensor(BODY26FK_MAP, dtype=torch.long)

    def forward(self, *args, root_trans=None, root_scale=None, orig_joints=False, **kwargs):
        out = super().forward(*args, **kwargs)                                          # :294-295
        if orig_joints:
            joints = out.joints[:, :24]
        else:
            extra = vertices2joints(self.J_regressor_extra, out.vertices)                # :299
            joints = torch.cat([out.joints, extra], dim=1)[:, self.joint_map, :]         # :300-301
        verts = out.vertices
        if root_trans is not None:                                                      # :309-315
            if root_scale is None:
                root_scale = torch.ones_like(root_trans[:, 0])
            pivot = joints[:, [0], :]
            verts = (verts - pivot) * root_scale[:, None, None] + root_trans[:, None, :]
            joints = (joints - pivot) * root_scale[:, None, None] + root_trans[:, None, :]
        return ModelOutput(vertices=verts, joints=joints, full_pose=out.full_pose, betas=out.betas,
                           global_orient=out.global_orient, body_pose=out.body_pose)

    def get_joints(self, betas=None, body_pose=None, global_orient=None, transl=None, root_trans=None, root_scale=None):
        """:318-343 -- forward kinematics only, rest joints regressed from the UNSHAPED template (betas ignored)."""
        pose = torch.cat([global_orient, body_pose], dim=1)
        B = pose.shape[0]
        J = torch.matmul(self.J_regressor, self.v_template).repeat((B, 1, 1))
        R = batch_rodrigues(pose.view(-1, 3)).view([B, -1, 3, 3])
        joints, _ = batch_rigid_transform(R, J, self.parents)
        if transl is not None:
            joints = joints + transl.unsqueeze(1)
        if root_trans is not None:
            if root_scale is None:
                root_scale = torch.ones_like(root_trans[:, 0])
            joints = (joints - joints[:, [0], :]) * root_scale[:, None, None] + root_trans[:, None, :]
        return joints
