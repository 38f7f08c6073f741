#!/bin/bash
# Like tools/build_variant.sh, for a flag that touches several sources:  tools/build_variant_multi.sh <name> "<flags>" a.hip b.hip ...
set -e
name=$1; extra=$2; shift 2
root=$(cd "$(dirname "$0")/.." && pwd)
csrc=$root/score_based_channels_amd/csrc
flags=$(make -C $csrc -pn 2>/dev/null | sed -n 's/^CXXFLAGS = //p' | head -1 | sed 's/\$(ARCH)/gfx950/')
mkdir -p /tmp/var_$name $root/tools/var
cp $csrc/build/*.o /tmp/var_$name/
for src in "$@"; do
  (cd $csrc && /opt/rocm/bin/hipcc $flags $extra -c $src -o /tmp/var_$name/${src%.hip}.o) &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC /tmp/var_$name/*.o -o $root/tools/var/libsbc_$name.so
python3 $root/tools/check_no_packed.py $root/tools/var/libsbc_$name.so
echo built tools/var/libsbc_$name.so
