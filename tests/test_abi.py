"""CPU-side checks of the product library: it builds for gfx950, loads without a GPU, and exports
every symbol ``include/densereg.h`` / ``include/densereg_profile.h`` declare and nothing else; the debug library adds
``include/densereg_debug.h`` (no compute calls)."""
import ctypes
import os
import re
import subprocess

import pytest

from tests.common import ROOT

from densereg_amd import _lib


def _ensure_built():
    if not os.path.exists(_lib.LIB_PATH):
        subprocess.check_call([os.path.join(ROOT, 'build.sh')], cwd=ROOT)


def _declared_symbols(headers):
    names = set()
    for hdr in headers:
        src = open(os.path.join(ROOT, 'include', hdr)).read()
        src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
        names |= set(re.findall(r'\b(dr_[a-z0-9_]+)\s*\(', src))
    return names


def _exported(path):
    out = subprocess.run(['nm', '-D', '--defined-only', path], capture_output=True, text=True, check=True).stdout
    return {ln.split()[-1] for ln in out.splitlines() if ln.split() and ln.split()[-1].startswith('dr_')}


def test_library_exports_every_declared_symbol():
    """The product library exports exactly include/densereg.h + include/densereg_profile.h -- and none of the dr_dbg_* test
    hooks; the debug library (same sources + -DDR_DEBUG_HOOKS) adds include/densereg_debug.h."""
    _ensure_built()
    lib = ctypes.CDLL(_lib.LIB_PATH)
    decl = _declared_symbols(('densereg.h', 'densereg_profile.h'))
    assert len(decl) >= 25
    for name in sorted(decl):
        assert hasattr(lib, name), 'libdensereg_hip.so does not export %s' % name
    assert decl == set(_lib.SIGNATURES), 'python binding and headers disagree: %s' % (decl ^ set(_lib.SIGNATURES))
    exported = _exported(_lib.LIB_PATH)
    assert exported == decl, 'product library exports beyond its headers: %s' % sorted(exported ^ decl)
    assert not [n for n in exported if n.startswith('dr_dbg_')]
    _lib.bind(lib)
    assert lib.dr_abi_version() == 1
    assert lib.dr_backend() == b'hip-gfx950'
    dbg_decl = _declared_symbols(('densereg_debug.h',))
    assert dbg_decl == set(_lib.DEBUG_SIGNATURES), dbg_decl ^ set(_lib.DEBUG_SIGNATURES)
    assert os.path.exists(_lib.DEBUG_LIB_PATH), 'build.sh builds the debug library next to the product'
    assert _exported(_lib.DEBUG_LIB_PATH) == decl | dbg_decl
    blob = open(_lib.LIB_PATH, 'rb').read()
    assert b'dr_dbg_conv_bench' not in blob and os.path.getsize(_lib.LIB_PATH) < os.path.getsize(_lib.DEBUG_LIB_PATH)


def test_library_contains_gfx950_code_object():
    _ensure_built()
    out = subprocess.run(['/opt/rocm/lib/llvm/bin/clang-offload-bundler', '--list', '--type=o', '--input=' + _lib.LIB_PATH],
                         capture_output=True, text=True)
    blob = open(_lib.LIB_PATH, 'rb').read()
    assert b'gfx950' in blob, out.stdout + out.stderr
    assert b'conv_igemm_kernel' in blob and b'vote_kernel' in blob


def test_product_loader_has_no_cpu_fallback():
    """Without a GPU dr_create must fail loudly (DR_E_DEVICE), not route anywhere else."""
    import pytest
    import torch
    if torch.cuda.is_available():
        pytest.skip('GPU present')
    _ensure_built()
    lib = _lib.load()
    with pytest.raises(_lib.DenseRegError) as e:
        _lib.Handle(lib, 1, 8, 2, 128, 3, 1, 0, False)
    assert e.value.code == -4 and 'no CPU fallback' in str(e.value)


def test_package_does_not_import_the_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, 'densereg_amd')):
        for f in files:
            if f.endswith('.py'):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r'^\s*(from|import)\s+oracle\b', src, flags=re.M), os.path.join(dirpath, f)
                assert 'hipemu' not in src and 'libdensereg_emu' not in src, os.path.join(dirpath, f)


def test_headers_are_c99_and_a_c_program_links_against_the_library(tmp_path):
    """include/*.h is a C ABI: it compiles as pedantic C99, and examples/abi_probe.c (host-side entry points only:
    no GPU needed) builds with gcc against libdensereg_hip.so and gets the right answers."""
    import shutil
    import subprocess
    gcc = shutil.which('gcc')
    if gcc is None:
        pytest.skip('no gcc in this environment')
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    inc, libdir = os.path.join(root, 'include'), os.path.join(root, 'densereg_amd', 'lib')
    both = tmp_path / 'both.c'
    both.write_text('#include "densereg.h"\n#include "densereg_profile.h"\n#include "densereg_debug.h"\nint main(void) { return 0; }\n')
    r = subprocess.run([gcc, '-std=c99', '-Wall', '-Wextra', '-pedantic', '-Werror', '-I', inc, '-fsyntax-only', str(both)],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    exe = str(tmp_path / 'abi_probe')
    r = subprocess.run([gcc, '-std=c99', '-Wall', '-Wextra', '-pedantic', '-I', inc, os.path.join(root, 'examples', 'abi_probe.c'), '-o', exe,
                        '-L', libdir, '-ldensereg_hip', '-Wl,-rpath,' + libdir, '-Wl,-rpath,/opt/rocm/lib'], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 0, (r.stdout, r.stderr)
    assert 'backend=hip-gfx950' in r.stdout and 'crc32c=e3069283' in r.stdout and 'unfilter_ok=1' in r.stdout
