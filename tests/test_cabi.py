"""CPU tests of the drop-in boundary: libff3d_hip.so builds for gfx950, loads, and exports exactly the
entry points include/ff3d.h declares (no compute calls - there is no GPU here)."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, 'include', 'ff3d.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(ff3d_[a-z0-9_]+)\s*\(', text)))


@pytest.fixture(scope='module')
def lib_path():
    from focalformer3d_amd import build
    return build.build(verbose=False)


def test_header_declares_the_hot_path_ops():
    syms = declared_symbols()
    for s in ('ff3d_msda_fwd', 'ff3d_msda_fused_fwd', 'ff3d_self_attention', 'ff3d_heatmap_nms', 'ff3d_topk', 'ff3d_query_gather',
              'ff3d_bev_flatten', 'ff3d_sine_embed', 'ff3d_roi_grid_sample', 'ff3d_box_decode', 'ff3d_cam_sample',
              'ff3d_nchw_to_nhwc'):
        assert s in syms


def test_library_exports_every_declared_symbol(lib_path):
    out = subprocess.run(['nm', '-D', '--defined-only', lib_path], capture_output=True, text=True, check=True).stdout
    exported = set(re.findall(r' T (ff3d_\w+)', out))
    assert exported == set(declared_symbols())


def test_ctypes_binding_matches_header(lib_path):
    from focalformer3d_amd import _lib
    assert sorted(_lib.SIGNATURES) == declared_symbols()
    lib = _lib.load()
    assert lib.ff3d_version() // 100 == 2
    assert lib.ff3d_status_string(0) == b'ok'
    assert lib.ff3d_topk_workspace_bytes(2, 1000) == (2 * 1000 + 2) * 8        # candidate keys + one counter per frame


def test_library_contains_gfx950_code_object(lib_path):
    data = open(lib_path, 'rb').read()
    assert b'gfx950' in data


def test_ops_refuse_cpu_tensors(lib_path):
    """The product path has no CPU fallback: CPU tensors raise instead of computing."""
    import torch
    from focalformer3d_amd import ops
    with pytest.raises(RuntimeError):
        ops.heatmap_nms(torch.zeros(1, 2, 4, 4))
    with pytest.raises(RuntimeError):
        ops.sine_embed(torch.zeros(4, 2), torch.ones(128), 1.0, 1.0)


def test_header_is_plain_c_and_links_from_c(lib_path, tmp_path):
    """The boundary is a C ABI: include/ff3d.h compiles as C99 with -Wall -Werror -pedantic (no C++-isms, no torch / HIP
    headers), and a C program that references every declared entry point links against libff3d_hip.so and runs (it calls only
    the two host-side queries - there is no GPU here)."""
    syms = declared_symbols()
    src = tmp_path / 'use_ff3d.c'
    refs = '\n'.join(f'  table[{i}] = (void (*)(void))&{s};' for i, s in enumerate(syms))
    src.write_text('#include <stdio.h>\n#include <string.h>\n#include "ff3d.h"\n'
                   'int main(void) {\n'
                   f'  void (*table[{len(syms)}])(void);\n{refs}\n'
                   f'  for (int i = 0; i < {len(syms)}; ++i) if (!table[i]) return 2;\n'
                   '  if (ff3d_version() / 100 != 2) return 3;\n'
                   '  if (strcmp(ff3d_status_string(0), "ok") != 0) return 4;\n'
                   '  printf("%d entry points\\n", (int)(sizeof table / sizeof table[0]));\n  return 0;\n}\n')
    exe = tmp_path / 'use_ff3d'
    inc = os.path.join(ROOT, 'include')
    subprocess.run(['gcc', '-std=c99', '-Wall', '-Werror', '-pedantic', '-fsyntax-only', '-I', inc, str(src)], check=True)
    libdir = os.path.dirname(lib_path)
    subprocess.run(['gcc', '-std=c99', '-Wall', '-Werror', '-I', inc, str(src), '-o', str(exe), '-L', libdir, '-lff3d_hip',
                    '-L/opt/rocm/lib', f'-Wl,-rpath,{libdir}', '-Wl,-rpath,/opt/rocm/lib'], check=True)
    out = subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout
    assert out.strip() == f'{len(syms)} entry points'
