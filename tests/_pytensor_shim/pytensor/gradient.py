from ._core import DisconnectedType, NullType, grad, grad_not_implemented  # noqa: F401
