"""kalign_amd -- MI355X (gfx950) implementation of Kalign's progressive-alignment hot path.

The product is the C-ABI library kalign_amd/libkalign_amd.so (include/kalign_amd.h), built
from kalign_amd/csrc by `make -C kalign_amd/csrc` (or __graft_entry__.build()).  This Python
package is the thin host-side mirror used by tests and bench.py; it never falls back to a
CPU implementation: without the HIP library (or without a GPU) calls raise.
"""
from .api import (Context, KalignAmdError, TaskRec, lib_path, load_library, msa_tree,  # noqa: F401
                  pairwise_batch)
