"""corticall_amd — MI355X-native linked de Bruijn graph traversal (one hot path of mcveanlab/Corticall).

The package is a thin host mirror of the reference's DeBruijnGraph / TraversalEngine API over
libldbg.so (hand-written HIP for gfx950, include/ldbg.h).  Importing it does not need a GPU;
opening a graph does.
"""
from ._native import (CortexJDKException, JavaNullPointerException, LdbgError, NativeLib,  # noqa: F401
                      NoSuchElementException, default_lib)
from .graph import CortexCollection, CortexGraph, CortexRecord  # noqa: F401
from .traversal import (AND, BOTH, FORWARD, OR, REVERSE, STOPPING_RULES, CortexLinks, CortexVertex, EnginePool,  # noqa: F401
                        TraversalEngine, TraversalEngineFactory, TraversalUtils, profile_get, profile_reset)
from .traversal import *  # noqa: F401,F403  (stopping-rule names: ContigStopper, DestinationStopper, ...)
from .partition import FindTips, Join, Partition, Sort  # noqa: F401,E402
from . import traversal_utils  # noqa: F401,E402
