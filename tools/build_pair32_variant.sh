#!/bin/bash
# Builds tools/var/libsbc_pair32.so: the product library + tools/experiments/conv_pair32.hip, with SBC_OP_CONV_PAIR launches of
# 32 channels / 16-pixel rows / conv_mode f16x2 routed to the experiment (-DSBC_WITH_PAIR32 in conv_pair.hip).
#   SBC_LIB_PATH=tools/var/libsbc_pair32.so python tools/prof_pair.py 1700       (SBC_PAIR32_NO_ILV=1: vector work outside the K loops)
# add -DSBC_PAIR_TIMING as first argument for the per-phase cycle sums (tools/prof_pair32_phases.py)
set -e
root=$(cd "$(dirname "$0")/.." && pwd)
csrc=$root/score_based_channels_amd/csrc
flags=$(make -C $csrc -pn 2>/dev/null | sed -n 's/^CXXFLAGS = //p' | head -1 | sed 's/\$(ARCH)/gfx950/')
mkdir -p /tmp/var_pair32 $root/tools/var
cp $csrc/build/*.o /tmp/var_pair32/
(cd $csrc && /opt/rocm/bin/hipcc $flags -DSBC_WITH_PAIR32 -c conv_pair.hip -o /tmp/var_pair32/conv_pair.o)
(cd $root/tools/experiments && /opt/rocm/bin/hipcc $flags -I$csrc "$@" -c conv_pair32.hip -o /tmp/var_pair32/conv_pair32.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC /tmp/var_pair32/*.o -o $root/tools/var/libsbc_pair32.so
echo built tools/var/libsbc_pair32.so
