"""MI355X-native lduMatrix solver hot path (OpenFOAM-2.2.x drop-in).

Loaded under the alias `openfoam_amd` (the directory name follows the reference
and is not an importable identifier; see __graft_entry__.py / tests/conftest.py).
"""
from . import cases, ldub  # noqa: F401
