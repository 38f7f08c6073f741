#!/bin/bash
# Builds tools/var/libsbc_wp.so = the product library + the conv_wp experiment (tools/experiments/conv_wp.hip) switched on for the
# 64 -> 64 layers of conv_mode f16x2.  [flags...] are added to conv_wp.hip (e.g. -DSBC_WP_TIMING).  SBC_NO_WP=1 switches it off at run time.
set -e
root=$(cd "$(dirname "$0")/.." && pwd)
csrc=$root/score_based_channels_amd/csrc
flags=$(make -C $csrc -pn 2>/dev/null | sed -n 's/^CXXFLAGS = //p' | head -1 | sed 's/\$(ARCH)/gfx950/')
mkdir -p /tmp/var_wp $root/tools/var
cp $csrc/build/*.o /tmp/var_wp/
(cd $csrc && /opt/rocm/bin/hipcc $flags -DSBC_WITH_WP -c conv_mfma.hip -o /tmp/var_wp/conv_mfma.o) &
(cd $csrc && /opt/rocm/bin/hipcc $flags "$@" -I$csrc -c $root/tools/experiments/conv_wp.hip -o /tmp/var_wp/conv_wp.o) &
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC /tmp/var_wp/*.o -o $root/tools/var/libsbc_wp.so
python3 $root/tools/check_no_packed.py $root/tools/var/libsbc_wp.so
echo built tools/var/libsbc_wp.so
