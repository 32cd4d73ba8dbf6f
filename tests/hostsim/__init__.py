"""TEST INFRASTRUCTURE: g++ builds of the templated kernel algorithms (host runtime, single thread) used by the CPU test
suite to check hand-written gradients and the optimiser's control flow without a GPU.  The product never loads these."""
import ctypes
import os
import subprocess
import tempfile
import zlib

HERE = os.path.dirname(os.path.abspath(__file__))
_CACHE = {}


def build(name, extra_flags=()):
    """Compiles tests/hostsim/<name>.cpp with g++ into a temp .so and loads it."""
    key = (name, tuple(extra_flags))
    if key in _CACHE:
        return _CACHE[key]
    # builds with extra flags (development probes) get a file of their own: they must never be picked up by the test-suite
    tag = '' if not extra_flags else '_%08x' % (zlib.crc32(' '.join(extra_flags).encode()) & 0xffffffff)
    out = os.path.join(tempfile.gettempdir(), 'glamr_hostsim_%s%s_%d.so' % (name, tag, os.getuid()))
    src = os.path.join(HERE, name + '.cpp')
    deps = [src] + [os.path.join(HERE, '..', '..', 'glamr_amd', 'csrc', f) for f in os.listdir(os.path.join(HERE, '..', '..', 'glamr_amd', 'csrc')) if f.endswith('.hpp')]
    if not os.path.exists(out) or any(os.path.getmtime(d) > os.path.getmtime(out) for d in deps):
        subprocess.check_call(['g++', '-O2', '-std=c++17', '-shared', '-fPIC', '-ffp-contract=off', src, '-o', out] + list(extra_flags))
    _CACHE[key] = ctypes.CDLL(out)
    return _CACHE[key]
