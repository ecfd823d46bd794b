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
if os.environ.get("LDBG_HOSTSIM_SO"):          # e.g. the AddressSanitizer build (make hostsim-asan), run once per round
    SO = os.path.abspath(os.environ["LDBG_HOSTSIM_SO"])
SO16 = os.path.join(ROOT, "tests", "hostsim", "_build16", "libldbg_hostsim.so")


def build(target="hostsim"):
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "corticall_amd", "csrc"), target, "-j8"],
                          stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)


def load(rebuild=True):
    """rebuild=False: workers spawned by a test that has already built the library (two makes at once would race)"""
    from corticall_amd import NativeLib
    if rebuild and not os.environ.get("LDBG_HOSTSIM_SO"):
        build()
    lib = NativeLib(SO)
    # one lane per wavefront unless the environment says otherwise — said explicitly: the two builds of the simulation used to share the
    # setting through a GNU-unique symbol (now -fno-gnu-unique), and a test that had switched 64 lanes on left it on for this one too
    lib.dll.ldbg_hostsim_set_lanes(int(os.environ.get("LDBG_HOSTSIM_LANES", "1")))
    return lib


def load_wavefront(lanes=64, rebuild=True):
    """the simulation with `lanes` lanes per wavefront in LOCK STEP (csrc/rt.h: one fibre per lane, wavefront primitives are barriers) and the
    device's 16 link-store elements per lane: the wave-cooperative code — lscoop.h (whole-wavefront and 16-lane group operations), table
    regrowth, path expansion, request bucketing — runs on the CPU as it does on the device"""
    from corticall_amd import NativeLib
    if rebuild:
        build("hostsim16")
    lib = NativeLib(SO16)
    lib.dll.ldbg_hostsim_set_lanes(int(lanes))
    return lib
