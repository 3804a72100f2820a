from .._core import Apply, Constant, Variable  # noqa: F401
