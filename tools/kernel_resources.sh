#!/bin/bash
# Per-kernel register / LDS / scratch usage of one translation unit, from the compiler's own remarks (no GPU needed).
#   tools/kernel_resources.sh render.hip [extra hipcc flags]
set -e
cd "$(dirname "$0")/../sigman_release_amd/csrc"
src=$1; shift
extra=""
case $src in preprocess.hip) extra="-ffp-contract=off";; render.hip) extra="-fno-slp-vectorize";; tile_sort.hip) extra="-fno-honor-nans";; esac
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics $extra "$@" -Rpass-analysis=kernel-resource-usage -c $src -o /dev/null 2>&1 |
  python3 -c "
import sys,re
cur=None;rows=[]
for l in sys.stdin:
    m=re.search(r'Function Name: (\S+)',l)
    if m: cur={'name':m.group(1)};rows.append(cur);continue
    m=re.search(r'remark: .*?\s+(VGPRs|AGPRs|SGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]|VGPR Spill|SGPR Spill): (\d+)',l)
    if m and cur is not None: cur[m.group(1).split(' ')[0]]=int(m.group(2))
import subprocess
for r in rows:
    n=subprocess.run(['c++filt',r['name']],capture_output=True,text=True).stdout.strip()
    n=re.sub(r'\(anonymous namespace\)::','',n); n=re.sub(r'\(.*','',n)
    print(f\"{n[:70]:70s} vgpr {r.get('VGPRs',0):4d} agpr {r.get('AGPRs',0):3d} sgpr {r.get('SGPRs',0):4d} scratch {r.get('ScratchSize',0):5d} occ {r.get('Occupancy',0):2d} lds {r.get('LDS',0):6d}\")
"
