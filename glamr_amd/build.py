"""Builds libglamr_hip.so (gfx950) in-tree with hipcc.  `python -m glamr_amd.build [--force]`."""
import glob
import os
import subprocess
import sys

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG_DIR, 'csrc')
LIB_PATH = os.path.join(PKG_DIR, 'libglamr_hip.so')
HEADER = os.path.join(os.path.dirname(PKG_DIR), 'include', 'glamr_hip.h')
HIPCC = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-Wall', '-Wno-unused-function']
# per-file flags.  grecon.hip: `/` and sqrtf as v_rcp / v_sqrt (+1 Newton step) instead of the correctly rounded ~10-instruction
# sequences -- the optimiser divides ~60 times per frame and iteration; results move in the last 1-2 ulp
# grecon.hip: approximate (2.5 ulp) division / sqrt for the few `/` left outside rotmath's rcp_/sqrt_; no SLP vectorisation -- packing
# pairs of fp32 operations (v_pk_fma_f32) costs as many register moves as it saves instructions and 36 more spilled registers
# (stage launch 42.3 -> 39.2 ms at 1024 scenes)
# init.hip: no fused multiply-adds -- init_data rounds like the reference's CPU operators (see rotmath.hpp, GLAMR_ROTMATH_IEEE)
FILE_FLAGS = {'grecon.hip': ['-fno-hip-fp32-correctly-rounded-divide-sqrt', '-fno-slp-vectorize'], 'init.hip': ['-ffp-contract=off']}
FILE_FLAGS['grecon_wide.hip'] = FILE_FLAGS['grecon.hip']          # (the same file, compiled for scenes of up to 32 persons)
INCLUDES = {'grecon_wide.hip': ['grecon.hip']}                        # sources that #include another source
EXTRA_FLAGS = {k: v.split() for k, v in (kv.split('=', 1) for kv in os.environ.get('GLAMR_EXTRA_FLAGS', '').split(';') if kv)}


def _sources():
    return sorted(glob.glob(os.path.join(CSRC, '*.hip')) + glob.glob(os.path.join(CSRC, '*.cpp')))


def _headers():
    return glob.glob(os.path.join(CSRC, '*.hpp')) + [HEADER]


def _stale():
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    return any(os.path.getmtime(p) > t for p in _sources() + _headers())


def build_library(force=False, verbose=False):
    """Compiles every source under csrc/ into one shared object.  Returns the library path."""
    if not force and not _stale():
        return LIB_PATH
    objs = []
    bdir = os.path.join(CSRC, 'build')
    os.makedirs(bdir, exist_ok=True)
    hdr_time = max(os.path.getmtime(p) for p in _headers())
    for src in _sources():
        obj = os.path.join(bdir, os.path.basename(src) + '.o')
        dep_time = max([os.path.getmtime(src), hdr_time] + [os.path.getmtime(os.path.join(CSRC, d)) for d in INCLUDES.get(os.path.basename(src), [])])
        if force or not os.path.exists(obj) or os.path.getmtime(obj) < dep_time:
            cmd = [HIPCC] + FLAGS + FILE_FLAGS.get(os.path.basename(src), []) + EXTRA_FLAGS.get(os.path.basename(src), []) + (['-x', 'hip'] if src.endswith('.cpp') else []) + ['-c', src, '-o', obj]
            if verbose:
                print(' '.join(cmd))
            subprocess.check_call(cmd)
        objs.append(obj)
    cmd = [HIPCC, '--offload-arch=gfx950', '-shared', '-fPIC'] + objs + ['-o', LIB_PATH]
    if verbose:
        print(' '.join(cmd))
    subprocess.check_call(cmd)
    return LIB_PATH


if __name__ == '__main__':
    print(build_library(force='--force' in sys.argv, verbose=True))
