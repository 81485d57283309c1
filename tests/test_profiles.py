"""profiles/ holds rocprofv3 summaries of the SHIPPED build: every kernel of this library named in the newest round's profiles
(`profiles/rNN_*_kernel_stats.csv`, `rNN_*_pmc.json`) must be an instantiation that HEAD's libearl_hip.so really holds (`nm -C`).  Round 5's kitchen profile still
named `kitchen_rollout_kernel<false>` after the kernel had become `<0|1|2>`: this test is what catches that.  No GPU needed."""
import csv
import glob
import json
import os
import re
import subprocess

from conftest import REPO

PROF = os.path.join(REPO, 'profiles')
LIB = os.path.join(REPO, 'earl_benchmark_amd', 'csrc', 'libearl_hip.so')
OURS = re.compile(r'^(void )?(earl::|\(anonymous namespace\)::)\w*kernel')      # kernels of this library: the name STARTS in its namespaces (torch's own kernels in the same traces are not ours to check)


def newest_round():
  tags = sorted({m.group(1) for f in os.listdir(PROF) for m in [re.match(r'(r\d\d)_.*_kernel_stats\.csv$', f)] if m})
  return tags[-1]


def library_kernels():
  out = subprocess.run(['nm', '-C', LIB], capture_output=True, text=True, check=True).stdout
  return {ln.split(' ', 2)[2].strip() for ln in out.splitlines() if len(ln.split(' ', 2)) == 3}


def norm(name):
  return re.sub(r'^void ', '', name.replace(' [clone .kd]', '')).replace(' ', '')


def test_every_profiled_kernel_of_the_newest_round_is_an_instantiation_of_heads_library():
  tag = newest_round()
  have = {norm(k) for k in library_kernels()}
  checked = 0
  for f in sorted(glob.glob(os.path.join(PROF, f'{tag}_*_kernel_stats.csv'))):
    for row in csv.DictReader(open(f)):
      if OURS.search(row['Name']):
        assert norm(row['Name']) in have, f'{os.path.basename(f)} names {row["Name"]!r}: not in libearl_hip.so (profile of an older build?)'
        checked += 1
  for f in sorted(glob.glob(os.path.join(PROF, f'{tag}_*_pmc.json'))):
    k = json.load(open(f)).get('kernel')
    if isinstance(k, str) and OURS.search(k):
      assert norm(k) in have, f'{os.path.basename(f)} names {k!r}: not in libearl_hip.so'
      checked += 1
  assert checked >= 5, f'{tag}: only {checked} kernels of this library found in the profiles'


def test_traffic_json_points_at_profiles_that_exist():
  tj = json.load(open(os.path.join(PROF, 'traffic.json')))
  for key, v in tj.items():
    src = v.get('source')
    assert src and os.path.exists(os.path.join(REPO, src)), (key, src)
  tag = newest_round()
  for w in ('sawyer_door', 'sawyer_peg', 'kitchen', 'minitaur'):           # the bench's static counter blocks come from the newest round's profiles
    assert tj[w]['source'].startswith(f'profiles/{tag}_'), (w, tj[w]['source'], tag)
