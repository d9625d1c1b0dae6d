"""`monotonic_align.monotonic_align.core.maximum_path_c` on the HIP path
(reference: monotonic_align/core.pyx:38-45)."""
from ...maxpath import maximum_path_c  # noqa: F401
