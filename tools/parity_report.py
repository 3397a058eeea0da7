"""Measured step-parity errors vs the fp64 oracle (GPU box): python tools/parity_report.py  -- prints the worst
normalised-max / relative-L2 gradient error, the median L2, moment and weight errors for a few configurations, without
asserting (the bounds live in tests/parity.py).  Run with MUNIT_DEBUG_NO_WINOGRAD=1 for the direct-convolution numbers."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.parity import run_step_parity
for gs, it, size in ((1, 3, 64), (0, 1, 64), (1, 1, 128)):
    rep = run_step_parity(size=size, batch=2, gen_state=gs, iters=it, device="cuda:0", check=False)
    print(gs, it, size, {k: rep.get(k) for k in ("loss_rel", "grad_nerr", "grad_l2", "grad_l2_median", "moment_l2",
                                                  "weight_abs", "weight_l2")}, rep["grad_kinks"], flush=True)
