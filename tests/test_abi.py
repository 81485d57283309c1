"""CPU-side checks of the drop-in boundary: the HIP library loads without a GPU and exports every symbol that
include/earl_tabletop.h declares; the ctypes view matches the header; the product refuses to run without a GPU."""
import ctypes as C
import os
import re

import pytest

from conftest import REPO, has_gpu
from earl_benchmark_amd import _abi

HEADER = os.path.join(REPO, 'include', 'earl_tabletop.h')
HEADERS = [HEADER, os.path.join(REPO, 'include', 'earl_glue.h'), os.path.join(REPO, 'include', 'earl_physics.h')]


def declared_functions():
  src = ''.join(open(h).read() for h in HEADERS)
  src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
  return re.findall(r'^\s*(?:int|const char\*)\s+(earl_\w+)\s*\(', src, flags=re.M)


HOST_ONLY = {'earl_host_set_threads', 'earl_host_version', 'earl_host_last_error'}      # libearl_host.so's own library functions


def is_host(name):
  return name.endswith('_cpu') or name in HOST_ONLY


def test_header_declares_what_the_binding_expects():
  names = declared_functions()
  assert len(names) == len(set(names)) >= 20
  dev = {n for n in names if not is_host(n)}
  assert dev == set(_abi.SIGNATURES), dev ^ set(_abi.SIGNATURES)
  assert {n for n in names if n.endswith('_cpu')} == set(_abi.HOST_SIGNATURES)            # the `_cpu` twins: csrc/libearl_host.so


def test_library_loads_and_exports_every_declared_symbol():
  lib = _abi.load()
  host = C.CDLL(_abi.HOST_LIB_PATH)
  for name in declared_functions():
    assert hasattr(host if is_host(name) else lib, name), name
    assert not hasattr(lib if is_host(name) else host, name), f'{name} is exported by the wrong library'
  assert lib.earl_version().startswith(b'earl-hip')


def test_struct_layouts_match_the_header():
  # sizes/offsets implied by the C declarations (int32 x10 + uint64 x2; 9 pointers; 5 pointers)
  assert C.sizeof(_abi.TabletopCfg) == 56 and _abi.TabletopCfg.seed.offset == 40 and _abi.TabletopCfg.counter.offset == 48
  assert C.sizeof(_abi.TabletopState) == 72 and C.sizeof(_abi.TabletopOut) == 40
  src = open(HEADER).read()
  cfg_fields = re.findall(r'^\s*(?:u?int\d+_t)\s+(\w+);', src[src.index('typedef struct earl_tabletop_cfg'):src.index('} earl_tabletop_cfg')], flags=re.M)
  assert cfg_fields == [f[0] for f in _abi.TabletopCfg._fields_]
  st_fields = re.findall(r'^\s*(?:const\s+)?\w+\*\s+(\w+);', src[src.index('typedef struct earl_tabletop_state'):src.index('} earl_tabletop_state')], flags=re.M)
  assert st_fields == [f[0] for f in _abi.TabletopState._fields_]
  out_fields = re.findall(r'^\s*\w+\*\s+(\w+);', src[src.index('typedef struct earl_tabletop_out'):src.index('} earl_tabletop_out')], flags=re.M)
  assert out_fields == [f[0] for f in _abi.TabletopOut._fields_]


def test_every_struct_layout_matches_what_gcc_sees(tmp_path):
  """compile a probe against the public headers (plain C, as a foreign-language binding would) and compare sizeof /
  offsetof of every struct with the ctypes mirrors"""
  import subprocess
  from earl_benchmark_amd.physics import CollisionModelStruct, LinkModelStruct, LinkModelStruct24
  mirrors = {'earl_tabletop_cfg': _abi.TabletopCfg, 'earl_tabletop_state': _abi.TabletopState, 'earl_tabletop_out': _abi.TabletopOut,
             'earl_motor_params': _abi.MotorParams, 'earl_kitchen_params': _abi.KitchenParams, 'earl_link_model': LinkModelStruct, 'earl_link_model24': LinkModelStruct24, 'earl_collision_model': CollisionModelStruct, 'earl_sawyer_cfg': _abi.SawyerCfg,
             'earl_sawyer_state': _abi.SawyerState, 'earl_sawyer_out': _abi.SawyerOut, 'earl_kitchen_cfg': _abi.KitchenCfg,
             'earl_kitchen_state': _abi.KitchenState, 'earl_kitchen_out': _abi.KitchenOut}
  lines = []
  for cname, cls in mirrors.items():
    lines.append(f'printf("{cname} %zu", sizeof({cname}));')
    lines += [f'printf(" %zu", offsetof({cname}, {f[0]}));' for f in cls._fields_]
    lines.append('printf("\\n");')
  src = ('#include <stdio.h>\n#include <stddef.h>\n#include "earl_tabletop.h"\n#include "earl_glue.h"\n#include "earl_physics.h"\n'
         'int main(void) {\n' + '\n'.join(lines) + '\nreturn 0; }\n')
  c = tmp_path / 'probe.c'
  c.write_text(src)
  exe = tmp_path / 'probe'
  subprocess.run(['gcc', '-std=c99', '-Wall', '-Werror', '-I', os.path.join(REPO, 'include'), '-o', str(exe), str(c)], check=True)
  out = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.strip().splitlines()
  assert len(out) == len(mirrors)
  for line, (cname, cls) in zip(out, mirrors.items()):
    tok = line.split()
    assert tok[0] == cname and int(tok[1]) == C.sizeof(cls), (cname, tok[1], C.sizeof(cls))
    assert [int(x) for x in tok[2:]] == [getattr(cls, f[0]).offset for f in cls._fields_], cname


def test_argument_validation_needs_no_gpu():
  lib = _abi.load()
  assert lib.earl_tabletop_step(None, None, None, None, None, None) == -1
  assert b'NULL' in lib.earl_last_error()
  assert lib.earl_tabletop_reward(-1, None, 0, 0, None, None, None) == -1


@pytest.mark.skipif(has_gpu(), reason='this container check is for GPU-less hosts')
def test_product_fails_loudly_without_a_gpu():
  from earl_benchmark_amd.envs import tabletop
  with pytest.raises(_abi.EarlHipError):
    tabletop.TabletopManipulation(num_envs=4)
  with pytest.raises(_abi.EarlHipError):
    tabletop.TabletopManipulation(num_envs=4, device='cuda:0')
  with pytest.raises(_abi.EarlHipError):
    tabletop.TabletopManipulation(num_envs=4, device='meta')
  # device='cpu' is a build of its own that a caller asks for by name (include/earl_tabletop.h `_cpu` entry points, tests/test_host_build.py): never a fallback
  import earl_benchmark_amd
  loader = earl_benchmark_amd.EARLEnvs('tabletop_manipulation', num_envs=8)     # tables / demos need no GPU; the envs do
  with pytest.raises(_abi.EarlHipError):
    loader.get_envs()


def test_product_never_imports_the_oracle():
  """oracle/ is test infrastructure: nothing under earl_benchmark_amd/ may reference it."""
  for root, _, files in os.walk(os.path.join(REPO, 'earl_benchmark_amd')):
    for f in files:
      if f.endswith(('.py', '.hip', '.h', '.cpp')):
        txt = open(os.path.join(root, f)).read()
        for line in txt.splitlines():
          if re.search(r'^\s*(from|import)\s+oracle\b|libearl_oracle|#include\s+"[^"]*oracle', line):
            raise AssertionError(f'{f}: {line}')


def test_importing_the_product_does_not_load_the_oracle():
  """the runtime counterpart of the source check above: importing the package and its env / physics / glue / wrapper modules in a
  fresh interpreter leaves no `oracle` module in sys.modules"""
  import subprocess
  import sys
  code = ('import sys, earl_benchmark_amd, earl_benchmark_amd.glue, earl_benchmark_amd.physics, earl_benchmark_amd.envs.sawyer_door, '
          'earl_benchmark_amd.envs.sawyer_peg, earl_benchmark_amd.wrappers; '
          'bad = [m for m in sys.modules if m == "oracle" or m.startswith("oracle.")]; print(bad); sys.exit(1 if bad else 0)')
  r = subprocess.run([sys.executable, '-c', code], cwd=REPO, capture_output=True, text=True)
  assert r.returncode == 0, r.stdout + r.stderr
