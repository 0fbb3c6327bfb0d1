#!/bin/sh
# Development aid: build an alternative libhefx.so with extra -D flags for hefx_keyswitch.hip into build/
# (travels to the GPU box; select with HEFX_LIB=build/libhefx_<tag>.so).   usage: tools/build_variant.sh tag -DX=1 ...
set -e
tag=$1; shift
root=$(cd "$(dirname "$0")/.." && pwd)
c=$root/seal_fyp_logistic_regression_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -mllvm -pragma-unroll-threshold=1048576 -x hip "$@" -c $c/hefx_keyswitch.hip -o $root/build/ks_$tag.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $root/build/libhefx_$tag.so $c/hefx_kernels.o $root/build/ks_$tag.o $c/hefx_encode.o $c/hefx_sample.o $c/hefx_capi.o
echo $root/build/libhefx_$tag.so
