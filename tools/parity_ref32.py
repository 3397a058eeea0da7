"""Is a step-gradient deviation from the fp64 oracle the kernels' or the problem's?  (GPU box)  python tools/parity_ref32.py [iters]
Per iteration: the HIP trainer's gradients AND the oracle's own fp32 evaluation (torch CPU, direct convolutions, same weights,
inputs and pinned ReLU / L1 branches) against the fp64 oracle, worst tensors first."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.parity import run_step_parity
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 3
rep = run_step_parity(size=64, batch=2, gen_state=1, iters=iters, device="cuda:0", check=False, ref32=True)
for it, rows in enumerate(rep["ref32"]):
    hip_med = sorted(r[2] for r in rows)[len(rows) // 2]
    r32_med = sorted(r[4] for r in rows)[len(rows) // 2]
    print("--- iteration %d: %d tensors; median L2  HIP %.2e   oracle-fp32 %.2e;  worst max  HIP %.2e   oracle-fp32 %.2e" % (
        it, len(rows), hip_med, r32_med, max(r[1] for r in rows), max(r[3] for r in rows)), flush=True)
    for r in sorted(rows, key=lambda r: -max(r[1], r[3]))[:8]:
        print("   %-56s HIP max %.2e l2 %.2e | oracle-fp32 max %.2e l2 %.2e" % (r[0][:56], r[1], r[2], r[3], r[4]))
