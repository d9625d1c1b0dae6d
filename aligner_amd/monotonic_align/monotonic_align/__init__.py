"""Inner package of the import-path shim (the reference keeps its compiled
`core` extension here: monotonic_align/__init__.py:3)."""
