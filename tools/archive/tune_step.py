"""Times earl_tabletop_step (one launch per env step) vs N through the C ABI with preallocated buffers (GPU box)."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, torch
import hip_harness as hx
from earl_benchmark_amd import _abi
lib = _abi.load()
sizes = [int(x) for x in (sys.argv[1].split(',') if len(sys.argv) > 1 else '4096,65536,262144,1048576,4194304,16777216'.split(','))]
for n in sizes:
  h = hx.HipTabletop(n, horizon=10**9)
  h.reset()
  act = (torch.rand(n, 3, device='cuda') * 2 - 1).contiguous()
  arrs, out = h._outs((n,))
  st = h._state()
  stream = torch.cuda.current_stream().cuda_stream
  reps = max(20, min(2000, int(4e9 // (n * 145))))
  for _ in range(5):
    lib.earl_tabletop_step(C.byref(h.cfg), C.byref(st), act.data_ptr(), None, C.byref(out), stream)
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  torch.cuda.synchronize(); e0.record()
  for _ in range(reps):
    lib.earl_tabletop_step(C.byref(h.cfg), C.byref(st), act.data_ptr(), None, C.byref(out), stream)
  e1.record(); torch.cuda.synchronize()
  us = e0.elapsed_time(e1) * 1e3 / reps
  print(f'step n={n:9d} us_per_launch={us:9.2f} GB/s={n * 145 / us / 1e3:8.1f} Gsteps/s={n / us / 1e3:7.2f}', flush=True)
  del h, act, arrs
  torch.cuda.empty_cache()
