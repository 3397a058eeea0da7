#!/bin/bash
# A/B two builds of libmunit_hip.so in ONE GPU session (box-to-box variation is larger than most kernel
# changes): A = munit_amd/libmunit_hip.so, B = $1 (default munit_amd/libmunit_hip_alt.so).
# Usage (GPU box): bash tools/ab_bench.sh [alt.so] [steps]
ALT=${1:-munit_amd/libmunit_hip_alt.so}
STEPS=${2:-5}
set -e
for round in 1 2; do
  echo "== A ($round)"; timeout -k 10 300 python tools/time_conv.py
  timeout -k 10 300 python bench.py --steps $STEPS --warmup 2 --no-cpu-baseline --no-roofline --no-modes | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('A ms_per_step', d['ms_per_step'])"
  echo "== B ($round)"; MUNIT_HIP_LIB=$PWD/$ALT timeout -k 10 300 python tools/time_conv.py
  MUNIT_HIP_LIB=$PWD/$ALT timeout -k 10 300 python bench.py --steps $STEPS --warmup 2 --no-cpu-baseline --no-roofline --no-modes | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('B ms_per_step', d['ms_per_step'])"
done
