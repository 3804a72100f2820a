from .._core import FunctionGraph  # noqa: F401
