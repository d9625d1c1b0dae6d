"""Import-path shim: `monotonic_align.maximum_path` on the MI355X HIP path.

Mirrors the reference package layout (monotonic_align/__init__.py:3,6) so a
caller can switch by putting aligner_amd/ on sys.path or calling
aligner_amd.install_dropin(); the implementation lives in aligner_amd.maxpath.
"""
from ..maxpath import maximum_path  # noqa: F401
from .monotonic_align.core import maximum_path_c  # noqa: F401
