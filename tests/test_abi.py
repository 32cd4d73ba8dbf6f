"""CPU-side checks of the drop-in boundary: the shared library loads and exports exactly the entry points the header declares."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, 'include', 'glamr_hip.h')).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    return sorted(set(re.findall(r'\b(glamr_[a-z0-9_]+)\s*\(', src)))


def test_header_and_binding_agree():
    from glamr_amd import _lib
    assert _declared() == _lib.exported_symbols()


def test_library_builds_and_exports_every_declared_symbol():
    from glamr_amd import build, _lib
    build.build_library()
    L = _lib.lib()
    for name in _declared():
        assert hasattr(L, name), name
    assert L.glamr_version() >= 100
    assert isinstance(L.glamr_last_error(), bytes)


def test_struct_sizes_match_the_header():
    """ctypes mirrors of the ABI structs must have the C layout (checked against a gcc-compiled probe)."""
    import subprocess
    import tempfile
    from glamr_amd import _lib
    code = '#include <stdio.h>\n#include "glamr_hip.h"\nint main(){printf("%zu %zu %zu %zu %zu %zu\\n", sizeof(glamr_tensor_desc), ' \
           'sizeof(glamr_stage_desc), sizeof(glamr_scene_batch), sizeof(glamr_param_layout), sizeof(glamr_raw_batch), ' \
           'sizeof(glamr_person_arrays));return 0;}\n'
    with tempfile.TemporaryDirectory() as d:
        open(os.path.join(d, 'p.c'), 'w').write(code)
        subprocess.check_call(['gcc', '-I', os.path.join(ROOT, 'include'), os.path.join(d, 'p.c'), '-o', os.path.join(d, 'p')])
        sizes = [int(x) for x in subprocess.check_output([os.path.join(d, 'p')]).split()]
    assert sizes == [ctypes.sizeof(_lib.TensorDesc), ctypes.sizeof(_lib.StageDesc), ctypes.sizeof(_lib.SceneBatch), ctypes.sizeof(_lib.ParamLayout),
                     ctypes.sizeof(_lib.RawBatch), ctypes.sizeof(_lib.PersonArrays)]


def test_product_never_imports_the_oracle():
    """The product package must not route through the checker (oracle/) or any CPU fallback."""
    bad = []
    for dirpath, _, files in os.walk(os.path.join(ROOT, 'glamr_amd')):
        for f in files:
            if f.endswith('.py'):
                txt = open(os.path.join(dirpath, f)).read()
                if re.search(r'^\s*(from|import)\s+oracle\b', txt, flags=re.M):
                    bad.append(os.path.join(dirpath, f))
    assert not bad, bad
