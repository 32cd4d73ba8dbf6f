"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the third-party `smplx` body-model arithmetic.

The reference imports its SMPL maths from the un-vendored, un-pinned package `smplx`
(`/root/reference/requirements.txt:1`; call sites `lib/models/smpl.py:7-8,295,299,329,332`).  `smplx` is not
installed here and cannot be, so this file restates the *published* algorithm (Loper et al., "SMPL: A Skinned
Multi-Person Linear Model", SIGGRAPH Asia 2015, eqs. 2-10, as implemented by the public smplx package's
`lbs.py` / `body_models.py` / `vertex_joint_selector.py`):

    v_shaped = v_template + shapedirs . betas
    J        = J_regressor @ v_shaped
    R        = rodrigues(pose)            (smplx variant: angle = || r + 1e-8 ||)
    v_posed  = v_shaped + posedirs^T . vec(R[1:] - I)
    A        = kinematic chain of [R | J_rel], made relative to the rest joints
    verts    = sum_j W[v, j] * A_j * [v_posed; 1]
    joints45 = [chain joints (24) ; 21 picked vertices]

Parity status: *unpinned* -- there is no smplx golden vector anywhere under /root/reference.
The restatement is anchored on the reference's own call sites and on closed-form identities checked in
tests/test_oracle_golden.py::test_smplx_closed_form_identities (rest pose returns the template; a pure root rotation rotates the mesh rigidly;
the posed joints equal the forward-kinematics chain).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import torch

# kinematic tree of the SMPL skeleton (public model constant, kintree_table[0])
SMPL_PARENTS = [-1, 0, 0, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 9, 9, 12, 13, 14, 16, 17, 18, 19, 20, 21]

# vertex ids that smplx appends to the 24 chain joints for the SMPL topology (public `vertex_ids.py`, 'smplh' table),
# in the order VertexJointSelector emits them: face (nose, reye, leye, rear, lear), feet (L big/small toe, L heel,
# R big/small toe, R heel), then finger tips left (thumb..pinky) and right.  Consistent with JOINT_MAP indices
# 24..44 at /root/reference/lib/models/smpl.py:35-57.
SMPL_EXTRA_VERTEX_IDS = [332, 6260, 2800, 4071, 583,
                         3216, 3226, 3387, 6617, 6624, 6787,
                         2746, 2319, 2445, 2556, 2673,
                         6191, 5782, 5905, 6016, 6133]


def blend_shapes(betas, shape_disps):
    """(B,L) x (V,3,L) -> (B,V,3) per-vertex displacement."""
    return torch.einsum('bl,mkl->bmk', betas, shape_disps)


def vertices2joints(J_regressor, vertices):
    """(J,V) x (B,V,3) -> (B,J,3)."""
    return torch.einsum('bik,ji->bjk', vertices, J_regressor)


def batch_rodrigues(rot_vecs, epsilon=1e-8):
    """(N,3) axis-angle -> (N,3,3).  smplx adds 1e-8 to every component *before* taking the norm."""
    n = rot_vecs.shape[0]
    angle = torch.norm(rot_vecs + 1e-8, dim=1, keepdim=True)
    rot_dir = rot_vecs / angle
    cos = torch.unsqueeze(torch.cos(angle), dim=1)
    sin = torch.unsqueeze(torch.sin(angle), dim=1)
    rx, ry, rz = torch.split(rot_dir, 1, dim=1)
    zeros = torch.zeros((n, 1), dtype=rot_vecs.dtype, device=rot_vecs.device)
    K = torch.cat([zeros, -rz, ry, rz, zeros, -rx, -ry, rx, zeros], dim=1).view((n, 3, 3))
    ident = torch.eye(3, dtype=rot_vecs.dtype, device=rot_vecs.device).unsqueeze(dim=0)
    return ident + sin * K + (1 - cos) * torch.bmm(K, K)


def _transform_mat(R, t):
    return torch.cat([torch.nn.functional.pad(R, [0, 0, 0, 1]),
                      torch.nn.functional.pad(t, [0, 0, 0, 1], value=1)], dim=2)


def batch_rigid_transform(rot_mats, joints, parents, dtype=torch.float32):
    """Kinematic chain.  rot_mats (B,J,3,3), joints (B,J,3), parents (J,) -> posed joints (B,J,3), A (B,J,4,4)."""
    joints = torch.unsqueeze(joints, dim=-1)
    rel_joints = joints.clone()
    rel_joints[:, 1:] -= joints[:, parents[1:]]
    transforms_mat = _transform_mat(rot_mats.reshape(-1, 3, 3), rel_joints.reshape(-1, 3, 1)).reshape(-1, joints.shape[1], 4, 4)
    chain = [transforms_mat[:, 0]]
    for i in range(1, parents.shape[0]):
        chain.append(torch.matmul(chain[parents[i]], transforms_mat[:, i]))
    transforms = torch.stack(chain, dim=1)
    posed_joints = transforms[:, :, :3, 3]
    joints_homogen = torch.nn.functional.pad(joints, [0, 0, 0, 1])
    rel_transforms = transforms - torch.nn.functional.pad(torch.matmul(transforms, joints_homogen), [3, 0, 0, 0, 0, 0, 0, 0])
    return posed_joints, rel_transforms


def lbs(betas, pose, v_template, shapedirs, posedirs, J_regressor, parents, lbs_weights, pose2rot=True):
    """Linear blend skinning.  Returns (verts (B,V,3), posed chain joints (B,J,3))."""
    batch_size = max(betas.shape[0], pose.shape[0])
    device, dtype = betas.device, betas.dtype
    v_shaped = v_template + blend_shapes(betas, shapedirs)
    J = vertices2joints(J_regressor, v_shaped)
    ident = torch.eye(3, dtype=dtype, device=device)
    if pose2rot:
        rot_mats = batch_rodrigues(pose.view(-1, 3)).view([batch_size, -1, 3, 3])
        pose_feature = (rot_mats[:, 1:, :, :] - ident).view([batch_size, -1])
        pose_offsets = torch.matmul(pose_feature, posedirs).view(batch_size, -1, 3)
    else:
        pose_feature = pose[:, 1:].view(batch_size, -1, 3, 3) - ident
        rot_mats = pose.view(batch_size, -1, 3, 3)
        pose_offsets = torch.matmul(pose_feature.view(batch_size, -1), posedirs).view(batch_size, -1, 3)
    v_posed = pose_offsets + v_shaped
    J_transformed, A = batch_rigid_transform(rot_mats, J, parents, dtype=dtype)
    W = lbs_weights.unsqueeze(dim=0).expand([batch_size, -1, -1])
    num_joints = J_regressor.shape[0]
    T = torch.matmul(W, A.view(batch_size, num_joints, 16)).view(batch_size, -1, 4, 4)
    homogen_coord = torch.ones([batch_size, v_posed.shape[1], 1], dtype=dtype, device=device)
    v_posed_homo = torch.cat([v_posed, homogen_coord], dim=2)
    v_homo = torch.matmul(T, torch.unsqueeze(v_posed_homo, dim=-1))
    return v_homo[:, :, :3, 0], J_transformed


class SMPLOutput:
    """Attribute bag mirroring smplx.utils.SMPLOutput (only the fields lib/models/smpl.py:303-308 reads)."""

    def __init__(self, vertices=None, joints=None, full_pose=None, global_orient=None, body_pose=None, betas=None):
        self.vertices, self.joints, self.full_pose = vertices, joints, full_pose
        self.global_orient, self.body_pose, self.betas = global_orient, body_pose, betas


class SMPLLayer(torch.nn.Module):
    """Restatement of `smplx.SMPL` for the arguments the reference passes (lib/models/smpl.py:277-296):
    SMPL(model_dir, pose_type=..., create_transl=False[, gender=, batch_size=]).forward(global_orient=, body_pose=,
    betas=, return_full_pose=, get_skin=).  `model_dir` must hold `SMPL_NEUTRAL.pkl` (any gender falls back to it) with the
    public SMPL model keys: v_template, shapedirs, posedirs, J_regressor, weights, kintree_table, f."""

    NUM_BODY_JOINTS = 23

    def __init__(self, model_path, *args, gender='neutral', num_betas=10, **kwargs):
        super().__init__()
        import os
        import pickle
        import numpy as np
        path = model_path
        if os.path.isdir(path):
            cand = os.path.join(path, 'SMPL_%s.pkl' % gender.upper())
            path = cand if os.path.exists(cand) else os.path.join(path, 'SMPL_NEUTRAL.pkl')
        with open(path, 'rb') as f:
            md = pickle.load(f, encoding='latin1')

        def arr(x):
            return np.asarray(x.todense() if hasattr(x, 'todense') else x)

        f32 = lambda x: torch.tensor(arr(x), dtype=torch.float32)
        shapedirs = f32(md['shapedirs'])[:, :, :num_betas]
        self.faces = arr(md['f']).astype(np.int64)
        self.register_buffer('faces_tensor', torch.tensor(self.faces, dtype=torch.long))
        self.register_buffer('v_template', f32(md['v_template']))
        self.register_buffer('shapedirs', shapedirs)
        self.register_buffer('J_regressor', f32(md['J_regressor']))
        num_pose_basis = arr(md['posedirs']).shape[-1]
        self.register_buffer('posedirs', f32(md['posedirs']).reshape(-1, num_pose_basis).T.contiguous())
        parents = torch.tensor(arr(md['kintree_table'])[0].astype(np.int64), dtype=torch.long)
        parents[0] = -1
        self.register_buffer('parents', parents)
        self.register_buffer('lbs_weights', f32(md['weights']))
        self.register_buffer('extra_joints_idxs', torch.tensor(SMPL_EXTRA_VERTEX_IDS, dtype=torch.long))

    def forward(self, betas=None, body_pose=None, global_orient=None, transl=None, return_verts=True,
                return_full_pose=False, pose2rot=True, **kwargs):
        full_pose = torch.cat([global_orient, body_pose], dim=1)
        batch_size = max(betas.shape[0], global_orient.shape[0], body_pose.shape[0])
        if betas.shape[0] != batch_size:
            betas = betas.expand(int(batch_size / betas.shape[0]), -1)
        vertices, joints = lbs(betas, full_pose, self.v_template, self.shapedirs, self.posedirs,
                               self.J_regressor, self.parents, self.lbs_weights, pose2rot=pose2rot)
        extra = torch.index_select(vertices, 1, self.extra_joints_idxs)
        joints = torch.cat([joints, extra], dim=1)
        if transl is not None:
            joints = joints + transl.unsqueeze(dim=1)
            vertices = vertices + transl.unsqueeze(dim=1)
        return SMPLOutput(vertices=vertices if return_verts else None, joints=joints, betas=betas,
                          global_orient=global_orient, body_pose=body_pose,
                          full_pose=full_pose if return_full_pose else None)
