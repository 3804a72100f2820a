"""TEST STAND-IN for the handful of PyTensor classes pymc_statespace_amd/op.py touches.

PyTensor / PyMC are installed neither in the build image nor on the GPU box (no network), so without this
the Op layer (make_node / perform / infer_shape / connection_pattern / L_op / the vectorize rule) would never
execute.  This package is NOT PyTensor and is never importable by the product: tests/conftest.py puts it on
sys.path only when the real library is absent, and only tests import it.  It implements just enough graph
machinery -- typed variables, Apply nodes, Op.__call__, a FunctionGraph with a clients map, a Python-loop
evaluator that calls Op.perform, reverse-mode grad() that calls Op.L_op with DisconnectedType placeholders,
and vectorize_graph() with the _vectorize_node single-dispatch hook -- with the real library's names and
calling conventions (pytensor 2.x: graph/op.py, graph/basic.py, gradient.py, graph/replace.py), so that
tests/test_pytensor_op.py runs unchanged against either.
"""
from __future__ import annotations

from functools import singledispatch

import numpy as np


# ---- types and variables -------------------------------------------------------------------------------
class Type:
    def __call__(self, name=None):
        return Variable(self, name=name)

    def __eq__(self, other):
        return type(self) is type(other) and self.__dict__ == other.__dict__

    def __hash__(self):
        return hash((type(self), tuple(sorted(self.__dict__.items()))))


class TensorType(Type):
    def __init__(self, dtype, ndim=None, shape=None):
        self.dtype = str(dtype)
        self.ndim = int(ndim if ndim is not None else len(shape))

    def __repr__(self):
        return f"TensorType({self.dtype}, ndim={self.ndim})"


class DisconnectedType(Type):
    """pytensor.gradient.DisconnectedType: the gradient of an output nothing downstream differentiates."""


class NullType(Type):
    def __init__(self, why=""):
        self.why = why


class Variable:
    def __init__(self, type, owner=None, index=None, name=None):
        self.type, self.owner, self.index, self.name = type, owner, index, name

    ndim = property(lambda self: self.type.ndim)
    dtype = property(lambda self: self.type.dtype)

    @property
    def shape(self):
        return _ShapeOf(self)

    def __mul__(self, other):
        return Mul()(self, other)

    __rmul__ = __mul__

    def __add__(self, other):
        return Add()(self, other)

    def sum(self):
        return SumAll()(self)

    def dimshuffle(self, *order):
        return DimShuffle(order)(self)

    def __repr__(self):
        return self.name or (f"{type(self.owner.op).__name__}.{self.index}" if self.owner else "<var>")


class Constant(Variable):
    def __init__(self, data, name=None):
        data = np.asarray(data)
        super().__init__(TensorType(data.dtype, data.ndim), name=name)
        self.data = data


class _ShapeOf:
    """x.shape: indexable / iterable like the real symbolic shape vector (each entry a scalar variable)."""

    def __init__(self, var):
        self.var = var

    def __getitem__(self, i):
        return Shape_i(i % self.var.ndim if self.var.ndim else 0)(self.var)

    def __iter__(self):
        return (self[i] for i in range(self.var.ndim))

    def __len__(self):
        return self.var.ndim


def as_tensor_variable(x, name=None, ndim=None):
    if isinstance(x, Variable):
        return x
    if isinstance(x, _ShapeOf):
        raise TypeError("stand-in: a whole symbolic shape is not a tensor here")
    return Constant(np.asarray(x), name=name)


class Apply:
    def __init__(self, op, inputs, outputs):
        self.op, self.inputs, self.outputs = op, list(inputs), list(outputs)
        for i, o in enumerate(self.outputs):
            o.owner, o.index = self, i

    nin = property(lambda self: len(self.inputs))
    nout = property(lambda self: len(self.outputs))


class Op:
    __props__ = ()
    default_output = None

    def make_node(self, *inputs):
        raise NotImplementedError

    def __call__(self, *inputs, **kwargs):
        node = self.make_node(*inputs, **kwargs)
        if self.default_output is not None:
            return node.outputs[self.default_output]
        return node.outputs[0] if len(node.outputs) == 1 else list(node.outputs)

    def _props(self):
        return tuple(getattr(self, p) for p in self.__props__)

    def __eq__(self, other):
        return type(self) is type(other) and self._props() == other._props()

    def __hash__(self):
        return hash((type(self), self._props()))

    def L_op(self, inputs, outputs, output_grads):
        return self.grad(inputs, output_grads)

    def connection_pattern(self, node):
        return [[True] * len(node.outputs) for _ in node.inputs]


# ---- the few tensor Ops the adapter's graphs contain ------------------------------------------------------
def _scalar_like(x):
    return TensorType("int64" if np.issubdtype(np.asarray(getattr(x, "data", 0)).dtype, np.integer) else "float64", 0)


class Shape_i(Op):
    __props__ = ("i",)

    def __init__(self, i):
        self.i = int(i)

    def make_node(self, x):
        return Apply(self, [as_tensor_variable(x)], [Variable(TensorType("int64", 0))])

    def perform(self, node, inputs, output_storage):
        output_storage[0][0] = np.asarray(inputs[0].shape[self.i], dtype=np.int64)

    def L_op(self, inputs, outputs, output_grads):
        return [DisconnectedType()()]

    def connection_pattern(self, node):
        return [[False]]


class Alloc(Op):
    """zeros(shape) / broadcast_to(x, shape): value followed by the shape entries."""
    __props__ = ("ndim",)

    def __init__(self, ndim):
        self.ndim = int(ndim)

    def make_node(self, value, *shape):
        ins = [as_tensor_variable(value)] + [as_tensor_variable(s) for s in shape]
        return Apply(self, ins, [Variable(TensorType("float64", self.ndim))])

    def perform(self, node, inputs, output_storage):
        shape = tuple(int(s) for s in inputs[1:])
        output_storage[0][0] = np.array(np.broadcast_to(np.asarray(inputs[0], dtype=np.float64), shape), order="C")

    def L_op(self, inputs, outputs, output_grads):
        # broadcast_to(x, (B,) + x.shape): the cotangent of x is the sum over the leading axes the broadcast added
        # (same-rank broadcasting of size-1 axes is not needed by the adapter's graphs)
        extra = self.ndim - inputs[0].ndim
        gx = SumLeading(extra)(output_grads[0]) if extra else output_grads[0]
        return [gx] + [DisconnectedType()() for _ in inputs[1:]]

    def connection_pattern(self, node):
        return [[True]] + [[False] for _ in node.inputs[1:]]


class _Elemwise(Op):
    def make_node(self, a, b):
        a, b = as_tensor_variable(a), as_tensor_variable(b)
        return Apply(self, [a, b], [Variable(TensorType("float64", max(a.ndim, b.ndim)))])


def _unbroadcast(g, like):
    """sum the cotangent over the axes numpy broadcasting added (symbolic): only the scalar * tensor case is needed."""
    return g if like.ndim == g.ndim else SumAll()(g)


class Mul(_Elemwise):
    def perform(self, node, inputs, output_storage):
        output_storage[0][0] = np.asarray(inputs[0] * inputs[1], dtype=np.float64)

    def L_op(self, inputs, outputs, output_grads):
        (g,) = output_grads
        a, b = inputs
        return [_unbroadcast(g * b, a), _unbroadcast(g * a, b)]


class Add(_Elemwise):
    def perform(self, node, inputs, output_storage):
        output_storage[0][0] = np.asarray(inputs[0] + inputs[1], dtype=np.float64)

    def L_op(self, inputs, outputs, output_grads):
        (g,) = output_grads
        return [_unbroadcast(g, inputs[0]), _unbroadcast(g, inputs[1])]


class DimShuffle(Op):
    """x.dimshuffle(0, 'x', 'x'): permute the axes named by integers, insert a broadcastable axis for every 'x'."""
    __props__ = ("order",)

    def __init__(self, order):
        self.order = tuple(order)

    def make_node(self, x):
        x = as_tensor_variable(x)
        if sorted(o for o in self.order if o != "x") != list(range(x.ndim)):
            raise ValueError(f"dimshuffle{self.order} does not name every axis of a {x.ndim}-d input once")
        return Apply(self, [x], [Variable(TensorType("float64", len(self.order)))])

    def perform(self, node, inputs, output_storage):
        x = np.transpose(np.asarray(inputs[0]), [o for o in self.order if o != "x"])
        output_storage[0][0] = np.ascontiguousarray(x[tuple(None if o == "x" else slice(None) for o in self.order)])

    def L_op(self, inputs, outputs, output_grads):
        return [grad_not_implemented(self, 0, inputs[0], "stand-in: DimShuffle")]


class SumLeading(Op):
    """sum over the first k axes."""
    __props__ = ("k",)

    def __init__(self, k):
        self.k = int(k)

    def make_node(self, x):
        x = as_tensor_variable(x)
        return Apply(self, [x], [Variable(TensorType("float64", x.ndim - self.k))])

    def perform(self, node, inputs, output_storage):
        output_storage[0][0] = np.asarray(np.sum(inputs[0], axis=tuple(range(self.k))), dtype=np.float64)

    def L_op(self, inputs, outputs, output_grads):
        return [grad_not_implemented(self, 0, inputs[0], "stand-in: SumLeading")]


class SumAll(Op):
    def make_node(self, x):
        return Apply(self, [as_tensor_variable(x)], [Variable(TensorType("float64", 0))])

    def perform(self, node, inputs, output_storage):
        output_storage[0][0] = np.asarray(np.sum(inputs[0]), dtype=np.float64)

    def L_op(self, inputs, outputs, output_grads):
        return [output_grads[0] * ones_like(inputs[0])]


def zeros(shape, dtype="float64"):
    shape = tuple(shape) if isinstance(shape, (tuple, list)) else (shape,)
    return Alloc(len(shape))(0.0, *shape)


def broadcast_to(x, shape):
    shape = tuple(shape)
    return Alloc(len(shape))(x, *shape)


def zeros_like(x):
    x = as_tensor_variable(x)
    return Alloc(x.ndim)(0.0, *x.shape)


def ones_like(x):
    x = as_tensor_variable(x)
    return Alloc(x.ndim)(1.0, *x.shape)


def _tensor_ctor(ndim):
    def ctor(name=None, dtype="float64"):
        return Variable(TensorType(dtype, ndim), name=name)
    return ctor


scalar, vector, matrix, tensor3, tensor4 = (_tensor_ctor(k) for k in range(5))


# ---- graph traversal, FunctionGraph, evaluator ------------------------------------------------------------
def toposort_from(outputs):
    order, seen = [], set()

    def visit(v):
        nd = v.owner
        if nd is None or id(nd) in seen:
            return
        seen.add(id(nd))
        for i in nd.inputs:
            visit(i)
        order.append(nd)

    for o in outputs:
        visit(o)
    return order


class FunctionGraph:
    def __init__(self, inputs, outputs):
        self.inputs, self.outputs = list(inputs), list(outputs)
        self.apply_nodes = toposort_from(self.outputs)
        self.clients = {}
        for nd in self.apply_nodes:
            for o in nd.outputs:
                self.clients.setdefault(o, [])
            for k, i in enumerate(nd.inputs):
                self.clients.setdefault(i, []).append((nd, k))
        for k, o in enumerate(self.outputs):
            self.clients.setdefault(o, []).append(("output", k))

    def toposort(self):
        return list(self.apply_nodes)


class _Maker:
    def __init__(self, fgraph):
        self.fgraph = fgraph


class Function:
    def __init__(self, inputs, outputs, on_unused_input="raise"):
        self.single = not isinstance(outputs, (list, tuple))
        outs = [outputs] if self.single else list(outputs)
        self.maker = _Maker(FunctionGraph(inputs, outs))
        used = {id(i) for nd in self.maker.fgraph.apply_nodes for i in nd.inputs} | {id(o) for o in outs}
        if on_unused_input == "raise":
            for i in inputs:
                if id(i) not in used:
                    raise ValueError(f"unused input {i!r} (pass on_unused_input='ignore')")

    def __call__(self, *values):
        fg = self.maker.fgraph
        if len(values) != len(fg.inputs):
            raise TypeError(f"expected {len(fg.inputs)} inputs, got {len(values)}")
        env = {}
        for var, val in zip(fg.inputs, values):
            val = np.asarray(val, dtype=var.type.dtype)
            if val.ndim != var.type.ndim:
                raise TypeError(f"input {var!r}: expected {var.type.ndim} dimensions, got {val.ndim}")
            env[id(var)] = val

        def value(v):
            return v.data if isinstance(v, Constant) else env[id(v)]

        for nd in fg.apply_nodes:
            storage = [[None] for _ in nd.outputs]
            nd.op.perform(nd, [value(i) for i in nd.inputs], storage)
            if hasattr(nd.op, "infer_shape"):
                # the real library trusts infer_shape in shape-only rewrites: execute it and hold it to the computed shapes
                # (outputs an Op was told not to compute are zero-size placeholders and exempt)
                shp = nd.op.infer_shape(fg, nd, [tuple(np.shape(value(i))) for i in nd.inputs])
                want = getattr(nd.op, "want", None)
                for k, (o, s) in enumerate(zip(nd.outputs, storage)):
                    if want is not None and k < len(want) and not want[k]:
                        continue
                    if tuple(int(x) for x in shp[k]) != tuple(np.shape(s[0])):
                        raise AssertionError(f"{type(nd.op).__name__}.infer_shape says {tuple(shp[k])} for output {k}, "
                                             f"perform produced {np.shape(s[0])}")
            for o, s in zip(nd.outputs, storage):
                if s[0] is None:
                    raise RuntimeError(f"{type(nd.op).__name__}.perform left an output unset")
                got = np.asarray(s[0])
                if got.ndim != o.type.ndim:
                    raise TypeError(f"{type(nd.op).__name__}: output {o.index} has {got.ndim} dimensions, "
                                    f"its type says {o.type.ndim}")
                env[id(o)] = got
        res = [value(o) for o in fg.outputs]
        return res[0] if self.single else res


def function(inputs, outputs, on_unused_input="raise", **_):
    return Function(inputs, outputs, on_unused_input)


# ---- reverse mode -----------------------------------------------------------------------------------------
def grad_not_implemented(op, x_pos, x, comment=""):
    return Variable(NullType(f"{type(op).__name__} input {x_pos}: {comment}"))


def _is_disconnected(v):
    return isinstance(v.type, DisconnectedType)


def grad(cost, wrt, disconnected_inputs="raise"):
    single = not isinstance(wrt, (list, tuple))
    wrts = [wrt] if single else list(wrt)
    if cost.ndim != 0:
        raise TypeError("grad: cost must be a scalar")
    gmap = {id(cost): ones_like(cost)}
    for nd in reversed(toposort_from([cost])):
        ogs = [gmap.get(id(o)) for o in nd.outputs]
        if all(g is None for g in ogs):
            continue
        pattern = nd.op.connection_pattern(nd)
        ogs = [g if g is not None else DisconnectedType()() for g in ogs]
        igs = nd.op.L_op(nd.inputs, nd.outputs, ogs)
        if len(igs) != len(nd.inputs):
            raise ValueError(f"{type(nd.op).__name__}.L_op returned {len(igs)} gradients for {len(nd.inputs)} inputs")
        for k, (inp, g) in enumerate(zip(nd.inputs, igs)):
            connected = any(pattern[k][j] and not _is_disconnected(ogs[j]) for j in range(len(nd.outputs)))
            if _is_disconnected(g):
                continue
            if not connected:
                raise ValueError(f"{type(nd.op).__name__}: L_op returned a gradient for input {k}, which its "
                                 "connection_pattern declares disconnected from every differentiated output")
            if isinstance(g.type, NullType):
                gmap[id(inp)] = g
            elif id(inp) in gmap and not isinstance(gmap[id(inp)].type, NullType):
                gmap[id(inp)] = gmap[id(inp)] + g
            elif id(inp) not in gmap:
                gmap[id(inp)] = g
    out = []
    for w in wrts:
        g = gmap.get(id(w))
        if g is None:
            if disconnected_inputs == "raise":
                raise ValueError(f"grad: {w!r} is disconnected from the cost")
            g = zeros_like(w)
        elif isinstance(g.type, NullType):
            raise NotImplementedError(f"grad undefined: {g.type.why}")
        out.append(g)
    return out[0] if single else out


# ---- vectorize_graph ----------------------------------------------------------------------------------------
@singledispatch
def _vectorize_node(op, node, *batched_inputs):
    if all(n.type.ndim == o.type.ndim for n, o in zip(batched_inputs, node.inputs)):
        return op.make_node(*batched_inputs)
    raise NotImplementedError(f"stand-in: no vectorize rule for {type(op).__name__}")


@_vectorize_node.register(Shape_i)
def _vectorize_shape_i(op, node, x):
    # a core dimension of an input that gained leading batch axes: the same entry, counted from the new front (a scalar, unbatched)
    extra = x.type.ndim - node.inputs[0].type.ndim
    return Shape_i(op.i + extra).make_node(x)


def vectorize_node(node, *batched_inputs):
    return _vectorize_node(node.op, node, *batched_inputs)


def vectorize_graph(outputs, replace):
    single = not isinstance(outputs, (list, tuple))
    outs = [outputs] if single else list(outputs)
    memo = {id(k): v for k, v in replace.items()}

    def new(v):
        return memo.get(id(v), v)

    for nd in toposort_from(outs):
        ins = [new(i) for i in nd.inputs]
        if all(a is b for a, b in zip(ins, nd.inputs)):
            continue
        nn = vectorize_node(nd, *ins)
        for o, no in zip(nd.outputs, nn.outputs):
            memo[id(o)] = no
    res = [new(o) for o in outs]
    return res[0] if single else res
