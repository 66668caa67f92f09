export AFF=0,1,2,3
python -m pytest tests/test_gpu_parity.py -q -x -k "sort or forward_artefacts or golden" 2>&1 | tail -3
python tools/profile_per_view.py 2>&1 | grep -v amdgpu.ids
python bench.py --no-variants --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('c2', d['value'], d['ms_per_step'], d['kernel_ms_per_step'])"
