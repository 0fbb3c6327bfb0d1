#!/bin/sh
# Development aid: compile hefx_keyswitch.hip for ONE ring size with resource-usage remarks and print the table
#   tools/probe_ks.sh 14 [-DX=1 ...]
ln=$1; shift
root=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p $root/build
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -mllvm -pragma-unroll-threshold=1048576 -x hip -DHEFX_ONLY_LOGN=$ln "$@" -Rpass-analysis=kernel-resource-usage --save-temps=obj -c $root/seal_fyp_logistic_regression_amd/csrc/hefx_keyswitch.hip -o $root/build/ks_probe.o 2> $root/build/ks_probe.rpass
python3 $root/tools/rpass_summary.py $root/build/ks_probe.rpass
