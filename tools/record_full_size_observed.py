"""Dev tool: merge the per-config excursion counts a GPU test run left in gpurun_out/full_size_observed_<name>.json
(tests/test_gpu_parity.py::_check_against_observed, run with SIGMAN_RECORD_OBSERVED=1) into tests/golden/full_size_observed.json.
usage: python tools/record_full_size_observed.py [note]"""
import glob, json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
path = os.path.join(ROOT, "tests", "golden", "full_size_observed.json")
rec = json.load(open(path)) if os.path.exists(path) else {}
for f in sorted(glob.glob(os.path.join(ROOT, "gpurun_out", "full_size_observed_*.json"))):
    rec.update(json.load(open(f)))
rec["_note"] = ("per config: {output: [values beyond the north_star tolerance (1e-4 abs images, 1e-4*max|g| gradients), max error]} of the HIP path "
                "against oracle/gsplat_ref.c on an MI355X; deterministic for a given build. " + (sys.argv[1] if len(sys.argv) > 1 else ""))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
import importlib.util
_spec = importlib.util.spec_from_file_location("tgp", os.path.join(ROOT, "tests", "test_gpu_parity.py"))
_tgp = importlib.util.module_from_spec(_spec); _spec.loader.exec_module(_tgp)
rec["_csrc_sha16"] = _tgp.kernel_sources_sha16()          # the record is a ceiling for THESE kernel sources only
try:
    rec["_commit"] = subprocess.check_output(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], text=True).strip()
except Exception:
    pass
json.dump(rec, open(path, "w"), indent=1, sort_keys=True)
print(json.dumps(rec, indent=1, sort_keys=True))
