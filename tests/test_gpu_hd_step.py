"""GPU: one whole dis_update + gen_update at BASELINE.json config #4's resolution against the fp64 oracle.  The oracle needs
about two minutes of host time at 512x512, so this file is collected LAST (tests/conftest.py): every other parity test has
reported before it starts."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from tests.parity import run_step_parity  # noqa: E402

pytestmark = pytest.mark.gpu


def test_step_matches_oracle_at_512():
    """The same whole-step comparison at config #4's resolution (configs/config_HD.yaml: 512x512 crops; batch 1 of its 4 --
    tests/test_gpu_step.py::test_hd_batch_gradient_is_the_mean_of_the_per_sample_gradients ties batch 4 to batch 1): 128x128
    trunk, 256 -> 512 up-sampling layer, five discriminator maps per scale, all against the fp64 oracle."""
    rep = run_step_parity(size=512, batch=1, gen_state=1, iters=1, device="cuda:0")
    print({k: v for k, v in rep.items() if not isinstance(v, list)})
