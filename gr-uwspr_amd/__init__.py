"""gr-uwspr_amd -- MI355X (gfx950) implementation of gr-uwspr's coarse (FDR) and
fine (sync_and_demodulate) search path behind the C ABI in include/uwspr_hip.h.

    csrc/     hand-written HIP kernels K1..K5 + the C ABI + the host tail
    host/     C++ mirror of the gr::uwspr block API on top of the C ABI
    native.py build (hipcc, gfx950) + ctypes binding
    context.py thin Python handle used by tests/ and bench.py
    synth.py  seeded synthetic frame generator (test/bench input only)
    dist.py   frame sharding + candidate gather across ranks (RCCL/gloo)

The directory name has a hyphen; import it as `gr_uwspr_amd` (root shim).
"""
from . import native  # noqa: F401
from .native import UwsprError, build  # noqa: F401
from .context import (Context, FrameView, Pipe, host_threads, host_set_ranks, deinterleave, fano_decode, fano_encode, decode_candidate, decode_batch,  # noqa: F401
                      unpack_message, c2_read, frontend_design, FRONTEND_GRC, FRONTEND_COMPACT, host_alloc, host_free)
from . import synth  # noqa: F401
from .sweep import sweep_grid, sweep_grid_uniform  # noqa: F401
