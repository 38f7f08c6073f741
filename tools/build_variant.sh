#!/bin/bash
# Tuning aid: build tools/var/libsbc_<name>.so with extra hipcc flags for ONE source file (others reuse csrc/build/*.o).
# The base flags come from the product Makefile (CXXFLAGS, including the mandatory -fno-slp-vectorize), so a variant differs
# from the product only by the flags given here; the result is checked for packed-fp32 instructions like the product.
# usage: tools/build_variant.sh <name> <source.hip> [flags...]     then: SBC_LIB_PATH=tools/var/libsbc_<name>.so python ...
set -e
name=$1; src=$2; shift 2
root=$(cd "$(dirname "$0")/.." && pwd)
csrc=$root/score_based_channels_amd/csrc
flags=$(make -C $csrc -pn 2>/dev/null | sed -n 's/^CXXFLAGS = //p' | head -1 | sed 's/\$(ARCH)/gfx950/')
mkdir -p /tmp/var_$name $root/tools/var
cp $csrc/build/*.o /tmp/var_$name/
(cd $csrc && /opt/rocm/bin/hipcc $flags "$@" -c $src -o /tmp/var_$name/${src%.hip}.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC /tmp/var_$name/*.o -o $root/tools/var/libsbc_$name.so
python3 $root/tools/check_no_packed.py $root/tools/var/libsbc_$name.so
echo built tools/var/libsbc_$name.so
