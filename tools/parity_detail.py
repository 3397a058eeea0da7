import os, sys
sys.path.insert(0, ".")
import tests.parity as P
orig_add, orig_fin = P.GradCheck.add, P.GradCheck.finish
def add(self, name, mine, ref, check=True):
    if not hasattr(self, "rows"): self.rows = []
    self.rows.append((P.l2err(mine, ref), P.nerr(mine, ref), name, float(ref.abs().max())))
    return orig_add(self, name, mine, ref, check)
def fin(self, check=True):
    rows = sorted(self.rows)
    print("--- iteration: median %.2e" % rows[len(rows)//2][0])
    for r in rows[::max(1, len(rows)//25)]: print("   l2 %.2e max %.2e gmax %.2e %s" % (r[0], r[1], r[3], r[2]))
    return orig_fin(self, check)
P.GradCheck.add, P.GradCheck.finish = add, fin
rep = P.run_step_parity(size=64, batch=2, gen_state=1, iters=3, device="cuda:0", check=False)
