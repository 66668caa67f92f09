set -x
O=gpurun_out/r2g; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -q -x --timeout=1200 > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -15 $O/pytest.log
AFF=0,1,2,3 python tools/profile_per_view.py 2>&1 | grep -v amdgpu.ids
AFF=0,1,2,3 SIGMAN_PY_NODE=1 python tools/profile_per_view.py 2>&1 | grep -v amdgpu.ids
