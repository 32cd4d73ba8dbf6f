"""Hand-written reverse-mode derivatives (glamr_amd/csrc/rotmath.hpp) vs torch autograd of the oracle formulation, on the host."""
import ctypes
import numpy as np
import pytest
import torch

from oracle.port import transforms as tf
from oracle.smplx_lbs import batch_rodrigues
from tests import hostsim


def _call(lib, name, x, gout, nout):
    x = np.ascontiguousarray(x, dtype=np.float32)
    gout = np.ascontiguousarray(gout, dtype=np.float32)
    out = np.zeros((x.shape[0], nout), dtype=np.float32)
    gx = np.zeros_like(x)
    fn = getattr(lib, name)
    fn.argtypes = [ctypes.c_int] + [ctypes.c_void_p] * 4
    fn(x.shape[0], x.ctypes.data, gout.ctypes.data, out.ctypes.data, gx.ctypes.data)
    return out, gx


def _rand_rot(n, rng):
    aa = torch.tensor(rng.normal(size=(n, 3)), dtype=torch.float32)
    aa[: n // 4] *= 2.5            # large angles: exercise every rotmat_to_quat branch
    return tf.aa_to_rotmat(aa).reshape(n, 9)


CASES = {
    't_rot6d_to_rotmat': (6, 9, lambda x: tf.sixd_to_rotmat(x).reshape(-1, 9), 'd6'),
    't_rotmat_to_quat': (9, 4, lambda x: tf.rotmat_to_quat(x.view(-1, 3, 3)), 'rot'),
    't_quat_to_aa': (4, 3, tf.quat_to_aa, 'quat'),
    't_aa_to_quat': (3, 4, tf.aa_to_quat, 'aa'),
    't_aa_to_rotmat_k': (3, 9, lambda x: tf.aa_to_rotmat(x).reshape(-1, 9), 'aa'),
    't_aa_to_rotmat_s': (3, 9, lambda x: batch_rodrigues(x).reshape(-1, 9), 'aa_s'),
    't_rotmat_to_aa': (9, 3, lambda x: tf.rotmat_to_aa(x.view(-1, 3, 3)), 'rot'),
    't_quat_mul': (8, 4, lambda x: tf.quat_mul(x[:, :4], x[:, 4:]), 'quat2'),
    't_atan2s': (2, 1, lambda x: tf.safe_atan2(x[:, 0], x[:, 1]).unsqueeze(-1), 'xy'),
    't_normalize3': (3, 3, tf.unit, 'vec'),
}


def _inputs(kind, rng, n=400):
    if kind == 'd6':
        x = rng.normal(size=(n, 6))
        x[0] = [1, 0, 0, 0, 1, 0]
    elif kind == 'rot':
        x = _rand_rot(n, rng).numpy() + rng.normal(size=(n, 9)) * 1e-3      # slightly off-manifold like the reference's products
    elif kind == 'quat':
        x = rng.normal(size=(n, 4))
        x /= np.linalg.norm(x, axis=1, keepdims=True)
        x[0] = [1, 0, 0, 0]
        x[1] = [-0.3, 0.2, 0.1, 0.9]
    elif kind == 'aa':
        x = rng.normal(size=(n, 3))
        x[0] = 0.0
        x[1] = [1e-4, -2e-4, 1e-4]        # Taylor branch of the kornia converter
        x[2] = [3.0, 0.5, -0.2]
    elif kind == 'aa_s':                  # smplx Rodrigues has no small-angle branch: 1 - cos(1e-4) is rounding noise in fp32
        x = rng.normal(size=(n, 3))
        x[0] = 0.0
        x[1] = [1e-2, -2e-2, 1e-2]
    elif kind == 'quat2':
        x = rng.normal(size=(n, 8))
    elif kind == 'xy':
        x = rng.normal(size=(n, 2))
        x[0] = [1e-8, 1e-7]
        x[1] = [0.0, -1.0]
    else:
        x = rng.normal(size=(n, 3))
        x[0] = 0.0
    return x.astype(np.float32)


@pytest.mark.parametrize('name', sorted(CASES))
def test_forward_and_backward_match_autograd(name):
    lib = hostsim.build('rotmath_shim')
    nin, nout, fn, kind = CASES[name]
    rng = np.random.default_rng(sum(map(ord, name)) % 1000)          # (not hash(name): Python randomises string hashes per process -- one run in a few dozen drew an ill-conditioned input)
    x = _inputs(kind, rng)
    gout = rng.normal(size=(x.shape[0], nout)).astype(np.float32)
    out, gx = _call(lib, name, x, gout, nout)
    xt = torch.tensor(x, requires_grad=True)
    ref = fn(xt)
    ref.backward(torch.tensor(gout))
    np.testing.assert_allclose(out, ref.detach().numpy(), rtol=2e-5, atol=2e-6, err_msg=name + ' forward')
    g_ref = xt.grad.numpy()
    scale = np.maximum(1.0, np.abs(g_ref).max(axis=1, keepdims=True))
    assert np.abs((gx - g_ref) / scale).max() < 2e-4, name + ' backward'


def test_sincos_accuracy():
    """rm::sincos_ (the optimiser's heading sine / cosine: three-constant Cody-Waite reduction + minimax polynomials) against float64:
    below 2 ulp and 1e-7 absolute over the range a prefix sum of 300 wrapped heading increments can reach, and still 1e-6 at 1e5."""
    lib = hostsim.build('rotmath_shim')
    rng = np.random.default_rng(7)
    for lim, tol_ulp, tol_abs in ((3.2, 2.0, 1.2e-7), (1000.0, 2.0, 1.2e-7), (1e5, 4.0, 1e-6)):
        x = np.concatenate([rng.uniform(-lim, lim, 400000), np.arange(-40, 41) * (np.pi / 4), [0.0, -0.0, 1e-30, -1e-8]]).astype(np.float32)
        out = np.zeros((x.shape[0], 2), np.float32)
        lib.t_sincos(ctypes.c_int(x.shape[0]), x.ctypes.data_as(ctypes.c_void_p), out.ctypes.data_as(ctypes.c_void_p))
        ref = np.stack([np.sin(x.astype(np.float64)), np.cos(x.astype(np.float64))], axis=1)
        err = np.abs(out - ref)
        ulp = np.spacing(np.abs(ref).astype(np.float32)).astype(np.float64)
        big = np.abs(ref) > 1e-3                       # near a zero of the function an ulp is tiny: the absolute bound applies there
        assert err.max() < tol_abs, (lim, err.max())
        assert (err / ulp)[big].max() < tol_ulp, (lim, (err / ulp)[big].max())
        assert np.all(np.abs(out[:, 0] ** 2 + out[:, 1] ** 2 - 1.0) < 4e-7)
