"""GPU: the drop-in autodiff ops a maintainer adds to athena (scripts/integration_check/hip_duvenaud_gno_ops.f90), LINKED against
libathena_mp.so and RUN from Fortran through a minimal working tape with diffstruc's callback protocol (mini_tape.f90; diffstruc
itself is not in this image): result nodes, operand links, `pure` get_partial_*_val callbacks asked one at a time by grad_reverse.
Every leaf gradient against the op-granular entry points; ONE fused device pass and ONE hand-over per two-partial node."""
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_drop_in_ops_run_through_a_tape_from_fortran(dev):
    exe = os.path.join(ROOT, "scripts", "integration_check", "run_ops")
    if not os.path.exists(exe):
        pytest.fail("scripts/integration_check/run_ops is not built: __graft_entry__.build() compiles it (amdflang)")
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    assert "RUN_OPS_OK 7 7" in r.stdout, r.stdout[-500:]
