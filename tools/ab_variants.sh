#!/bin/bash
# Time tools/time_conv.py under several builds of the library in ONE GPU session (box-to-box variation is larger than most
# kernel changes): bash tools/ab_variants.sh name1 name2 ...   (munit_amd/libmunit_hip_<name>.so; "base" = the product library)
for round in 1 2; do
  for n in "$@"; do
    if [ "$n" = base ]; then unset MUNIT_HIP_LIB; else export MUNIT_HIP_LIB=$PWD/munit_amd/libmunit_hip_$n.so; fi
    echo "== $n ($round)"; timeout -k 10 200 python tools/time_conv.py || exit 1
  done
done
