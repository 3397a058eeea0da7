run() { python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-roofline --no-modes 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['ms_per_step'], d['timing']['ms_per_step_median'])"; }
for i in 1 2; do
run default
MUNIT_DEBUG_NO_WINOGRAD_WGRAD=1 run direct_wgrad
MUNIT_NO_SIDE_STREAM=1 run no_side_stream
MUNIT_NO_BRANCH_STREAMS=1 run no_branch_streams
MUNIT_NO_SIDE_STREAM=1 MUNIT_NO_BRANCH_STREAMS=1 run single_stream
done
