#!/bin/bash
# Build a variant of libsigman_gsplat.so into tools/ab/<name>.so: the tree's objects with ONE translation unit recompiled with extra flags.
#   tools/build_ab.sh <name> <unit.hip> [extra hipcc flags]          e.g.  tools/build_ab.sh w4 render.hip -DSGR_SEG_WAVES=4
set -e
cd "$(dirname "$0")/../sigman_release_amd/csrc"
name=$1; unit=$2; shift 2
make -s -j4 ../lib/libsigman_gsplat.so
extra=""
case $unit in preprocess.hip) extra="-ffp-contract=off";; render.hip) extra="-fno-slp-vectorize";; tile_sort.hip) extra="-fno-honor-nans";; esac
mkdir -p ../../tools/ab
o=/tmp/ab_${name}_${unit%.hip}.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics $extra "$@" -c $unit -o $o
objs=""
for f in api preprocess binning tile_sort render knn loss rasterize; do if [ "$f.hip" = "$unit" ]; then objs="$objs $o"; else objs="$objs $f.o"; fi; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/ab/$name.so $objs
echo built tools/ab/$name.so
