#!/bin/bash
# A/B of library variants on the GPU box: tools/ab.sh name1 name2 ...   ("base" = the product library)
cd "$(dirname "$0")/.."
for v in "$@"; do
  if [ "$v" = base ]; then unset SBC_LIB_PATH; else export SBC_LIB_PATH=$PWD/tools/var/libsbc_$v.so; fi
  echo "== $v"
  python tools/prof_conv.py 32 32 3 1 1700 64 16 --mode ${AB_MODE:-wx2} --flags 1 --iters 50 2>&1 | tail -1
  python tools/prof_conv.py 64 64 3 1 1700 32 8 --mode ${AB_MODE:-wx2} --flags 1 --iters 50 2>&1 | tail -1
  python tools/prof_conv.py 64 64 3 1 1700 16 4 --mode ${AB_MODE:-wx2} --flags 1 --iters 50 2>&1 | tail -1
  python tools/prof_conv.py 128 128 3 1 1700 8 2 --mode ${AB_MODE:-wx2} --flags 1 --iters 50 2>&1 | tail -1
done
