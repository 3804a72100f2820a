from .._core import _vectorize_node, vectorize_graph, vectorize_node  # noqa: F401
