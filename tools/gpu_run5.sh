export AFF=0,1,2,3
for g in 0 1; do GRAPHS=$g python tools/profile_per_view.py 2>&1 | grep -v amdgpu.ids; done
GRAPHS=0 PROF=1 python tools/profile_per_view.py 2>&1 | grep -v amdgpu.ids | head -60
