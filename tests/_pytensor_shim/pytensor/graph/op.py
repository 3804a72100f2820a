from .._core import Op  # noqa: F401
