"""pytensor.tensor names used by pymc_statespace_amd/op.py and tests/test_pytensor_op.py (stand-in, see _core.py)."""
from ._core import (TensorType, as_tensor_variable, broadcast_to, matrix, ones_like, scalar, tensor3, tensor4,  # noqa: F401
                    vector, zeros, zeros_like)
from ._core import grad  # noqa: F401
