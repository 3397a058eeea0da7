"""Golden vectors for ExtraAdam from the reference's scripts/extraadam.py (build container only).

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_extraadam.py

The reference file has no import statements, so importing it raises NameError;
its source is executed here in a namespace that supplies the three names it expects (Optimizer, torch,
math).  Output: tests/golden/golden_extraadam.npz = initial parameters, the gradient sequence, the call
sequence and the parameters after every call (float64)."""
import math
import os
import types
import warnings

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
MODES = ["extrapolation", "step", "extrapolation", "extrapolation", "step"]


def main():
    ns = types.ModuleType("ref_extraadam")
    ns.__dict__.update(dict(Optimizer=torch.optim.Optimizer, torch=torch, math=math))
    with open("/root/reference/scripts/extraadam.py") as f:
        exec(compile(f.read(), "extraadam.py", "exec"), ns.__dict__)
    warnings.filterwarnings("ignore")
    torch.manual_seed(5)
    ps = [torch.nn.Parameter(torch.randn(7, 5, dtype=torch.float64)),
          torch.nn.Parameter(torch.randn(11, dtype=torch.float64))]
    gseq = [[torch.randn_like(p) * 0.3 for p in ps] for _ in MODES]
    opt = ns.ExtraAdam(ps, lr=1e-3, betas=(0.5, 0.999), weight_decay=1e-4)
    flat = lambda ts: np.concatenate([t.detach().numpy().reshape(-1) for t in ts])
    arrays = {"p0": flat(ps), "g": np.stack([flat(gs) for gs in gseq]), "modes": np.array(MODES)}
    trace = []
    for k, mode in enumerate(MODES):
        for p, g in zip(ps, gseq[k]):
            p.grad = g.clone()
        getattr(opt, mode)()
        trace.append(flat(ps))
    arrays["trace"] = np.stack(trace)
    np.savez_compressed(os.path.join(HERE, "golden_extraadam.npz"), **arrays)
    print("wrote golden_extraadam.npz", arrays["trace"].shape)


if __name__ == "__main__":
    main()
