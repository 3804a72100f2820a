"""TEST STAND-IN, not PyTensor -- see _core.py.  Only tests/ puts this package on sys.path."""
from . import _core, gradient, tensor  # noqa: F401
from ._core import function  # noqa: F401
from .gradient import grad  # noqa: F401

__version__ = "0.0-standin"
IS_TEST_STANDIN = True
