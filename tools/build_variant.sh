#!/bin/bash
# Tuning aid: build tools/var/libsbc_<name>.so with extra hipcc flags for ONE source file (others reuse csrc/build/*.o).
# usage: tools/build_variant.sh <name> <source.hip> [flags...]     then: SBC_LIB_PATH=tools/var/libsbc_<name>.so python ...
set -e
name=$1; src=$2; shift 2
root=$(cd "$(dirname "$0")/.." && pwd)
csrc=$root/score_based_channels_amd/csrc
mkdir -p /tmp/var_$name $root/tools/var
cp $csrc/build/*.o /tmp/var_$name/
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function "$@" -c $csrc/$src -o /tmp/var_$name/${src%.hip}.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC /tmp/var_$name/*.o -o $root/tools/var/libsbc_$name.so
echo built tools/var/libsbc_$name.so
