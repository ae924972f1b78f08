"""Seeded runs of the randomised consistency checks under scratch/ (each a self-contained script: random shapes, sizes and
options for one kernel family, compared with the oracle, with NumPy, with the Python definitions or with another form of the
same kernel).  The long runs live in scratch/README.md; here a few cases per family run inside the test process."""
import os
import runpy
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("script,cases,seed", [
    ("eval_fuzz.py", 14, 11),        # fused evaluation: precision x prescan x hint kind x sliced give the same lists
    ("bpr_fuzz.py", 14, 12),         # BPR step: pull / atomic / deterministic forms agree; deterministic repeats bit for bit
    ("ngcf_fuzz.py", 10, 13),        # NGCF step vs oracle/ngcf.py
    ("cdae_fuzz.py", 12, 14),        # fused CDAE step (both decoders) vs the autograd route
    ("lists_fuzz.py", 20, 15),       # yr_cdae_train_lists vs the dense route's compaction and the sampling law
    ("sampler_fuzz.py", 20, 16),     # device triplet sampler: permutation, negatives, windows
    ("metrics_fuzz.py", 40, 17),     # device metrics vs metric.py
    ("topk_fuzz.py", 20, 18),        # masked row-wise top-k vs NumPy, bit-exact order
    ("optim_fuzz.py", 12, 19),       # dense Adam / AdamW / SGD vs oracle/adam.py
])
def test_randomised_consistency(device, script, cases, seed, monkeypatch, capsys):
    path = os.path.join(ROOT, "scratch", script)
    monkeypatch.setattr(sys, "argv", [path, str(cases), str(seed)])
    runpy.run_path(path, run_name="__main__")
    assert "cases agree" in capsys.readouterr().out
