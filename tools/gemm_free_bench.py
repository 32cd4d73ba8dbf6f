"""DEVELOPMENT AID (GPU): tile-shape / prefetch variants of gemm_free_kernel (tools/gemm_free_bench.hip) -- checked against torch,
timed alone on the priors' shapes and as a train of launches beside a resident optimiser stage."""
import ctypes
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import numpy as np
import torch

VARIANTS = {0: '32x64  PD1 w4 (product)', 1: '32x64  PD2 w3', 2: '32x128 PD1 w2', 3: '32x32  PD2 w4'}


def lib():
    so, src = os.path.join(HERE, os.environ.get('GLAMR_GFB_SO', '_gemm_free_bench.so')), os.path.join(HERE, 'gemm_free_bench.hip')
    deps = [src, os.path.join(HERE, '..', 'glamr_amd', 'csrc', 'nn_free.hpp'), os.path.join(HERE, '..', 'glamr_amd', 'csrc', 'nn_kernels.hpp')]
    if not os.path.exists(so) or any(os.path.getmtime(so) < os.path.getmtime(d) for d in deps):
        subprocess.check_call(['hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-shared', '-fPIC', src,
                               os.path.join(HERE, '..', 'glamr_amd', 'csrc', 'api_common.cpp'), '-o', so])
    L = ctypes.CDLL(so)
    L.gfb_launch.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int,
                             ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    return L


def to_frag(X):
    """row-major [M][ld] (M a multiple of 32, ld of 16) -> fragment-major (csrc/nn_free.hpp x32_off)"""
    M, ld = X.shape
    return X.reshape(M // 32, 32, ld // 16, 2, 8).permute(0, 2, 3, 1, 4).contiguous().reshape(M, ld)


def from_frag(Y):
    M, ld = Y.shape
    return Y.reshape(M // 32, ld // 16, 2, 32, 8).permute(0, 3, 1, 2, 4).contiguous().reshape(M, ld)


def planes(W):
    """[N][K] fp32 -> two fp16 planes in the fragment order of glamr_nets_create (nets.hip up_lin)"""
    N, K = W.shape
    Np = (N + 63) // 64 * 64
    Wp = np.zeros((Np, K), np.float32)
    Wp[:N] = W
    hi = Wp.astype(np.float16)
    lo = (Wp - hi.astype(np.float32)).astype(np.float16)
    n, k = np.meshgrid(np.arange(Np), np.arange(K), indexing='ij')
    dst = (((n // 32) * (K // 16) + k // 16) * 64 + (n % 32) + 32 * ((k % 16) // 8)) * 8 + k % 8
    out = np.zeros((2, Np * K), np.float16)
    out[0, dst.ravel()] = hi.ravel()
    out[1, dst.ravel()] = lo.ravel()
    return out, Np * K


def main():
    dev = torch.device('cuda:0')
    L = lib()
    g = torch.Generator().manual_seed(0)
    shapes = [(51200, 256, 256), (51200, 768, 256), (51200, 256, 512)]
    data = {}
    for (M, N, K) in shapes:
        X = (torch.randn(M, K, generator=g)).to(dev)
        W = (torch.randn(N, K, generator=g) * 0.05)
        b = torch.randn((N + 63) // 64 * 64, generator=g).to(dev)
        pl, plane = planes(W.numpy())
        Ws = torch.from_numpy(pl.view(np.int16).copy()).to(dev)
        Y = torch.empty(M, N, device=dev)
        ref = X[:2048].double() @ W.to(dev).double().t() + b[:N].double()
        data[(M, N, K)] = (X, to_frag(X), Ws, plane, b, Y, ref)
    st = torch.cuda.current_stream()

    def launch(v, key, frag=1, stream=None):
        X, Xf, Ws, plane, b, Y, _ = data[key]
        M, N, K = key
        rc = L.gfb_launch(v, frag, (Xf if frag else X).data_ptr(), K, Ws.data_ptr(), plane, b.data_ptr(), Y.data_ptr(), N, M, N, K, ctypes.c_void_p((stream or st).cuda_stream))
        assert rc == 0, rc

    def time_of(fn, reps=5):
        fn()
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(reps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            fn()
            e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1))
        return best

    for frag in (1, 0):
        for v, name in VARIANTS.items():
            line = '%-24s %s' % (name, 'fragment-major' if frag else 'row-major     ')
            for key in shapes:
                data[key][5].zero_()
                launch(v, key, frag)
                torch.cuda.synchronize()
                Y = from_frag(data[key][5]) if frag else data[key][5]
                err = (Y[:2048].double() - data[key][6]).abs().max().item()
                ms = time_of(lambda: launch(v, key, frag))
                line += ' | %s %.3f ms %5.0f TF/s err %.1e' % ('x'.join(map(str, key)), ms, 2.0 * key[0] * key[1] * key[2] / ms / 1e9, err)
            print(line, flush=True)

    if os.environ.get('GLAMR_GFB_NO_STAGE'):
        return
    # beside a resident optimiser stage
    from glamr_amd import _lib
    from glamr_amd.global_recon import packing
    from glamr_amd.global_recon.configs import get_config
    from glamr_amd.utils import synth
    from oracle.port import build
    from oracle import make_golden as mg
    from tests.grecon_common import j_local_from_oracle
    root = build.ensure_synthetic_assets('/tmp/glamr_bench_assets')
    cfg = get_config('glamr_dynamic')
    md = synth.make_smpl_model()
    in_dict = synth.make_in_dict(seed=0, num_frames=300, num_persons=1, smpl_model=md)
    ora = build.load_optimizer(root, cfg)
    d0 = ora.init_data(in_dict, latents=mg.latents_for(in_dict, 0))
    jl = j_local_from_oracle(ora.smpl, d0)
    GL = _lib.lib()
    packed = packing.PackedScenes([d0] * 1024, [jl] * 1024, dev)
    sd = packing.stage_desc(cfg['opt_stage_specs']['init_opt'], cfg['grecon_model_specs'], False, niters=500)
    sb = packed.struct()
    ws = torch.empty(GL.glamr_grecon_workspace_bytes(packed.S, packed.P, packed.T), dtype=torch.uint8, device=dev)
    s1, s2 = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)

    def stage():
        _lib.check(GL.glamr_grecon_run_stage(ctypes.byref(sb), ctypes.byref(sd), None, _lib.ptr(ws), ctypes.c_void_p(s1.cuda_stream)))

    def both(do_stage, fill):
        torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True)
        e0.record(torch.cuda.current_stream())
        s1.wait_event(e0)
        s2.wait_event(e0)
        if do_stage:
            stage()
        if fill:
            fill()
        e1, e2 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e1.record(s1)
        e2.record(s2)
        torch.cuda.synchronize()
        return e0.elapsed_time(e1), e0.elapsed_time(e2)

    stage()
    torch.cuda.synchronize()
    print('stage alone %.2f ms' % both(True, None)[0], flush=True)
    key = (51200, 768, 256)
    for v, name in VARIANTS.items():
        n = 150
        fill = lambda: [launch(v, key, 1, s2) for _ in range(n)]
        alone = min(both(False, fill)[1] for _ in range(2))
        t = both(True, fill)
        print('%-14s train of %d x %s: alone %.2f ms | beside the stage: stage %.2f train %.2f | sum %.2f' % (name, n, 'x'.join(map(str, key)), alone, t[0], t[1], 29.4 + alone), flush=True)


if __name__ == '__main__':
    main()
