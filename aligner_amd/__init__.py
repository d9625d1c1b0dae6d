"""aligner_amd -- MI355X-native TTS alignment hot path (drop-in for xiaozhah/Aligner's
monotonic_align.maximum_path and the soft-attention front end that feeds it).

    from aligner_amd import maximum_path, align, soft_attention
    import aligner_amd; aligner_amd.install_dropin()   # then: from monotonic_align import maximum_path

All compute is in libaligner_amd.so (hand-written HIP for gfx950, C ABI in
include/aligner_amd.h).  Importing this package does not load the library; the
first call does, and raises if it was not built or no GPU is visible.
"""
from __future__ import annotations

import sys

from .maxpath import Alignment, align, maximum_path, maximum_path_c, read_status  # noqa: F401
from .softattn import (AlignmentEncoderParams, alignment_encoder, conv1d, soft_attention)  # noqa: F401
from .objective import beta_binomial_prior, forward_sum, forward_sum_loss, regulate  # noqa: F401
from .mobo import BoundarySearch, boundary_search, boundary_search_backward, soft_boundaries  # noqa: F401

__all__ = ["Alignment", "align", "maximum_path", "maximum_path_c", "read_status",
           "soft_attention", "conv1d", "alignment_encoder", "AlignmentEncoderParams",
           "forward_sum", "forward_sum_loss", "beta_binomial_prior", "regulate", "boundary_search", "boundary_search_backward", "soft_boundaries", "BoundarySearch", "install_dropin"]


def install_dropin() -> None:
    """Register this package's shim as the top-level `monotonic_align` module, so
    `from monotonic_align import maximum_path` and
    `from monotonic_align.monotonic_align.core import maximum_path_c` (the two
    import paths of the reference, __init__.py:3,6) resolve to the HIP path."""
    from . import monotonic_align as shim
    from .monotonic_align import monotonic_align as inner
    from .monotonic_align.monotonic_align import core
    sys.modules["monotonic_align"] = shim
    sys.modules["monotonic_align.monotonic_align"] = inner
    sys.modules["monotonic_align.monotonic_align.core"] = core
