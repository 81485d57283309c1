#!/usr/bin/env python3
"""Local half of tools/profile_all.sh: gpurun_out/prof_* (merged back from the GPU box) -> profiles/<tag>_* and profiles/traffic.json, for all five bench workloads.
usage: python tools/summarize_all.py r06 [workload ...]"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
ws = sys.argv[2:] or ['tabletop', 'sawyer_door', 'sawyer_peg', 'kitchen', 'minitaur']
T = os.path.join(ROOT, 'tools')
for w in ws:
  cmd = {'tabletop': ['summarize_profiles.py', tag], 'sawyer_door': ['summarize_sawyer.py', w, tag], 'sawyer_peg': ['summarize_sawyer.py', w, tag],
         'kitchen': ['summarize_kitchen.py', tag, 'kitchen'], 'minitaur': ['summarize_kitchen.py', tag, 'minitaur']}[w]
  r = subprocess.run([sys.executable, os.path.join(T, cmd[0])] + cmd[1:], capture_output=True, text=True)
  print(w, 'rc', r.returncode, (r.stderr.strip().splitlines() or [''])[-1][:200])
