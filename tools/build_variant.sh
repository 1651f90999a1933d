#!/bin/bash
# build_variant.sh NAME [extra hipcc flags...] -> tools/_variants/libradixsort_hip_NAME.so (A/B and diagnostic builds;
# select at run time with RSX_LIB=<path>).  The directory is git-ignored but travels to the GPU box.
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
NAME=$1; shift
mkdir -p "$ROOT/tools/_variants"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-strict-aliasing -shared -fPIC -I"$ROOT/include" "$@" \
    "$ROOT/radix-sort_amd/csrc/rsx_capi.hip" -o "$ROOT/tools/_variants/libradixsort_hip_$NAME.so"
echo "built tools/_variants/libradixsort_hip_$NAME.so"
