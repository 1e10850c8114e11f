"""Error types at the drop-in boundary.

Same names and hierarchy as reference tidy3d/exceptions.py:6-56 so that user
code catching ``tidy3d.exceptions.SetupError`` etc. keeps working.  When the
real tidy3d package is importable its classes are re-used (so ``isinstance``
checks against tidy3d's own classes hold); otherwise local classes with the
same names are defined.
"""

try:  # pragma: no cover - exercised only on hosts that have tidy3d
    from tidy3d.exceptions import (  # type: ignore
        Tidy3dError,
        ValidationError,
        SetupError,
        DataError,
        Tidy3dNotImplementedError,
        Tidy3dKeyError,
    )
except Exception:  # tidy3d (or one of its dependencies) is not importable here

    class Tidy3dError(ValueError):
        """Any error in tidy3d (ref exceptions.py:6)."""

    class Tidy3dKeyError(Tidy3dError):
        """Could not find a key (ref exceptions.py:19)."""

    class ValidationError(Tidy3dError):
        """Error when constructing components (ref exceptions.py:23)."""

    class SetupError(Tidy3dError):
        """Error regarding the setup of the components (ref exceptions.py:27)."""

    class DataError(Tidy3dError):
        """Error accessing data (ref exceptions.py:43)."""

    class Tidy3dNotImplementedError(Tidy3dError):
        """A functionality is not (yet) supported (ref exceptions.py:51)."""


class SolverLibraryError(RuntimeError):
    """The HIP solver library is missing, failed to load, or returned an error.

    There is deliberately no CPU fallback: the product path fails loudly.
    """
