#!/usr/bin/env python3
"""the constraint Hessian H = M + J' D J of the steppers (nv = 23, two envs per wave, one wave per SIMD, all CUs busy) on the vector ALU vs on the fp64
matrix pipe (v_mfma_f64_16x16x4_f64), plus the instruction's raw issue rate and dependent latency.  tools/ubench/mfma_f64_hessian.hip; results -> DESIGN.md 17.4.
  python tools/bench_mfma_hessian.py [--build]"""
import ctypes as C
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC, LIB = os.path.join(ROOT, 'tools', 'ubench', 'mfma_f64_hessian.hip'), os.path.join(ROOT, 'tools', 'ubench', 'libmfma_hessian.so')
if '--build' in sys.argv or not os.path.exists(LIB):
  subprocess.run(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-ffp-contract=off', '-fPIC', '-shared', '-o', LIB, SRC], check=True)
  if '--build' in sys.argv:
    sys.exit(0)
import torch                                     # device memory only
lib = C.CDLL(LIB)
NP, KMAX, NV = 32, 44, 23
nwg = 256
nenv = nwg * 8
rng = np.random.default_rng(0)


def run(K, reps, mode):
  J = np.zeros((nenv, KMAX, NP)); J[:, :K, :NV] = rng.normal(size=(nenv, K, NV))
  D = np.zeros((nenv, KMAX)); D[:, :K] = rng.uniform(0.5, 2.0, size=(nenv, K))
  A = rng.normal(size=(nenv, NP, NP)); M = A @ A.transpose(0, 2, 1)
  tJ, tD, tM = (torch.from_numpy(x).cuda() for x in (J, D, M))
  tH = torch.zeros(nenv, NP, NP, dtype=torch.float64, device='cuda')
  cyc = torch.zeros(nwg * 4, dtype=torch.int64, device='cuda')
  rc = lib.run_hessian(C.c_void_p(tJ.data_ptr()), C.c_void_p(tD.data_ptr()), C.c_void_p(tM.data_ptr()), C.c_void_p(tH.data_ptr()), C.c_void_p(cyc.data_ptr()), nwg, K, reps, mode)
  assert rc == 0
  want = M + np.einsum('ekl,ek,ekm->elm', J, D, J)
  return tH.cpu().numpy(), want, cyc.cpu().numpy().astype(np.float64)


print(f'device: {torch.cuda.get_device_name(0)}; {nwg} workgroups x 4 waves (one per SIMD), two envs per wave, nv = {NV} (padded to {NP})')
for K in (8, 24, 40):
  for mode, name in ((0, 'VALU, lane = column'), (1, 'MFMA f64 16x16x4, 3 tiles per env')):
    H, want, _ = run(K, 1, mode)
    err = np.abs(H[:, :NV, :NV] - want[:, :NV, :NV]).max() / np.abs(want).max()
    _, _, c1 = run(K, 1, mode)
    _, _, c9 = run(K, 9, mode)
    per = (c9 - c1) / 8
    print(f'K = {K:2d} rows  {name:36s}: {np.median(per):8.0f} cycles per wave (two envs) median, p90 {np.percentile(per, 90):8.0f}; max rel. error vs numpy {err:.1e}')
_, _, c1 = run(40, 1, 2); _, _, c9 = run(40, 9, 2)
print(f'issue rate : {np.median(c9 - c1) / 8 / (40 * 8):6.1f} cycles per v_mfma_f64_16x16x4_f64 (eight independent accumulators, one wave per SIMD, all four SIMDs of every CU busy)')
_, _, c1 = run(40, 1, 3); _, _, c9 = run(40, 9, 3)
print(f'dependent  : {np.median(c9 - c1) / 8 / (40 * 8):6.1f} cycles per instruction in a chain through the accumulator')
