"""Nothing the kernels read may rely on what hipMalloc or LDS happen to hold.  The host simulation hands out zeroed "device" memory and its
simulated LDS starts as zeros, both kinder than the device; with LDBG_HOSTSIM_POISON=1 fresh device memory and the LDS of a starting
wavefront are filled with 0xAB, and with LDBG_HOSTSIM_LDS_CHECK=1 a link-store element of the LDS that is read before the running wavefront
wrote it aborts the process with a backtrace (csrc/rt.h, csrc/engine.h).  A few parity cases run under both, on the one-lane simulation
and in lock step at 64 lanes with the device's 16 link-store elements per lane — in a child process, the switches are read once.

(Found with this: the simulation's wave_fence() was no barrier, so in lock step a lane could read a link-store element before the fibre
of the lane that writes it had run — a soak seed failed only when an earlier test had left 64 lanes switched on; the device orders the two
by the lock step itself.)"""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import sys, pathlib, tempfile
sys.path.insert(0, %(root)r)
from oracle import pyoracle as orc
from tests import hostsim, parity_cases as pc
orc.lib()
lanes = int(sys.argv[1])
lib = hostsim.load(rebuild=False) if lanes == 1 else hostsim.load_wavefront(lanes, rebuild=False)
tmp = lambda: pathlib.Path(tempfile.mkdtemp(prefix="poison_"))
pc.case_run_steps(orc, lib, tmp(), 201)
pc.case_dense_cycles(orc, lib, tmp(), 2)
if lanes == 1:          # (the lock-step run keeps to the two cases above: a fibre switch per lane and primitive)
    pc.case_random_walks(orc, lib, tmp(), 31, 4, True)
    pc.case_dfs_dense(orc, lib, tmp(), 1)
    pc.case_dfs_run_steps(orc, lib, tmp(), 231)
    pc.case_facade(orc, lib, tmp(), 31, 3, True)
print("poison ok", lanes)
"""


@pytest.mark.timeout(1500)
@pytest.mark.parametrize("lanes", [1, 64])
def test_poisoned_memory_and_lds(orc, lanes):
    from tests import hostsim
    hostsim.build()
    if lanes > 1:
        hostsim.build("hostsim16")
    env = dict(os.environ, LDBG_HOSTSIM_POISON="1", LDBG_HOSTSIM_LDS_CHECK="1")
    r = subprocess.run([sys.executable, "-c", CHILD % {"root": ROOT}, str(lanes)], env=env, capture_output=True, text=True, timeout=1400)
    assert r.returncode == 0 and ("poison ok %d" % lanes) in r.stdout, (r.stdout[-1500:], r.stderr[-3000:])
