set -x
tools/micro/bts
bash tools/gpu_run2.sh
