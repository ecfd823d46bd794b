"""TEST-ONLY host simulation of libldbg.

`make -C corticall_amd/csrc hostsim` compiles the very same kernel sources as plain C++
(-DLDBG_HOSTSIM): a "launch" runs the kernel body once per simulated thread on the CPU.  The CPU-only
CI container uses it to check kernel logic against the oracle; it is not part of the product, is
never loaded by corticall_amd on its own, and the GPU parity tests (-m gpu) do not use it.
"""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
SO = os.path.join(ROOT, "tests", "hostsim", "_build", "libldbg_hostsim.so")


def build():
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "corticall_amd", "csrc"), "hostsim", "-j8"],
                          stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)


def load(rebuild=True):
    """rebuild=False: workers spawned by a test that has already built the library (two makes at once would race)"""
    from corticall_amd import NativeLib
    if rebuild:
        build()
    return NativeLib(SO)
