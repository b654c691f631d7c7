"""Shapes the reference accepts and the fast kernels do not cover (any total_seq_length / d_model / n_head: transformers4rec/config/
transformer.py:218-260 GPT-2, :432-482 XLNet, :493-534 BERT): two training steps and an inference call of the module mirror must run
through the general kernels (csrc/xlnet_attn_long.hip, the 16-word instance of mask_targets_kernel) with finite losses.  Numerical
parity of those kernels is in tests/test_kernels_gpu.py / test_e2e_gpu.py (reference fixtures at total_seq_length 100 / 150) and
tests/test_round6_gpu.py (dropout steps against the oracle); this file guards the COVERAGE (tools/shape_sweep.py is the full list)."""
import os
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))


@pytest.mark.parametrize("case", [
    ("xlnet", "mlm", 65, 128, 4), ("xlnet", "clm", 255, 32, 2), ("xlnet", "mlm", 500, 32, 2), ("xlnet", "mlm", 50, 256, 8),
    ("xlnet", "mlm", 20, 320, 2), ("xlnet", "mlm", 20, 100, 4), ("gpt2", "clm", 300, 32, 2), ("gpt2", "clm", 50, 192, 4),
    ("bert", "mlm", 200, 48, 2), ("bert", "mlm", 1023, 16, 1), ("xlnet", "mlm", 100, 64, 4, "concat"),
    ("gpt2", "clm", 150, 64, 2, "concat"),
], ids=lambda c: "-".join(str(x) for x in c))
def test_shape_trains_and_infers(case):
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import shape_sweep

    losses = shape_sweep.run(*case)
    assert len(losses) == 2 and all(l == l for l in losses)
