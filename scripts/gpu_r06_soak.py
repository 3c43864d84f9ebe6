"""soak of round 6's new device code on random inputs: the fuzz tests of tests/test_gpu_fuzz.py (Kipf on random block-diagonal
batches through whichever gather the library picks; the device graph builder's radix passes against the host builder) and the banded
Kipf test of tests/test_gpu_ops.py, called with seeds far beyond the ones pytest runs.  python scripts/gpu_r06_soak.py [n_seeds]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import pytest, torch
from oracle import oracle
import test_gpu_fuzz as fz

n = int(sys.argv[1]) if len(sys.argv) > 1 else 150
dev = torch.device("cuda:0")
torch.zeros(1, device=dev)
t0 = time.time(); ran = skipped = 0
for seed in range(100, 100 + n):
    for fn, args in ((fz.test_kipf_on_random_batches_fuzz, (dev, oracle, seed)), (fz.test_device_graph_builder_fuzz, (dev, seed))):
        try:
            fn(*args); ran += 1
        except pytest.skip.Exception:
            skipped += 1
print(f"R06_SOAK_OK {ran} draws clean, {skipped} degenerate, {time.time() - t0:.0f} s")
