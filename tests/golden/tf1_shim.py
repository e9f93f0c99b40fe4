"""A minimal TensorFlow-1 stand-in on torch autograd -- BUILD-CONTAINER TOOL of tests/golden/make_graph_golden.py, nothing else.

TensorFlow cannot be installed here, so the reference's graph code (core/tensorflow_state.py,
core/regularization_functions.py) never ran anywhere in this project.  make_graph_golden.py runs the reference's OWN text of those two
files (lib2to3 scratch copy outside the repo) against this module installed as `tensorflow` in the scratch directory: every op the two
files use is mapped to the torch op of the same meaning, EAGERLY (the graph-building code then simply computes), `function.Defun` becomes
a torch.autograd.Function whose backward is the reference's own grad_func, and `train.AdamOptimizer` is TF1's Adam update
(tensorflow/python/training/adam.py: lr_t = lr sqrt(1 - b2^t) / (1 - b1^t); var -= lr_t m / (sqrt(v) + eps)).

What this buys: the loop bounds, slices, signs and op order of a6-a14 in the golden vectors are the REFERENCE's, not a human restatement.
What it does not: it is a stand-in for TensorFlow, so by the letter of the rules the oracle stays "parity unpinned by the reference's own
runtime" (DESIGN.md section 2 says so).  `float32` is mapped to torch.float64 by default: those fixtures (graph_*.npz) check the algebra at
fp64.  With QOC_TF1_SHIM_FP32=1 in the environment at import `float32` IS torch.float32 (and complex64 torch.complex64): the reference's text
then runs at the reference's OWN precision, and the fixtures graph32_*.npz made that way bound the distance between the fp64 engine and what
the real reference computes (tier 2 of SURVEY.md 8c).
Nothing of this file or of the reference travels to the GPU box; only the numeric outputs do (tests/golden/graph_*.npz).
"""
import contextlib
import os
import types

import numpy as np
import torch

FP32 = os.environ.get('QOC_TF1_SHIM_FP32', '0') == '1'
float32 = torch.float32 if FP32 else torch.float64            # see the module docstring
float64 = torch.float64
complex64 = torch.complex64 if FP32 else torch.complex128
int32 = torch.int64

_VARIABLES = []                    # (tensor, trainable) in creation order


def _t(x, dtype=None):
    if isinstance(x, torch.Tensor):
        return x if dtype is None or x.dtype == dtype else x.to(dtype)
    return torch.as_tensor(np.asarray(x), dtype=dtype if dtype is not None else None)


def constant(value, dtype=None, name=None, shape=None):
    return _t(value, dtype).clone().detach()


def Variable(initial_value, trainable=True, dtype=None, name=None):
    v = _t(initial_value, dtype).clone().detach()
    v.requires_grad_(bool(trainable))
    _VARIABLES.append((v, bool(trainable)))
    return v


def ones(shape, dtype=float32, name=None):
    return torch.ones(tuple(int(s) for s in shape), dtype=dtype)


def zeros(shape, dtype=float32, name=None):
    return torch.zeros(tuple(int(s) for s in shape), dtype=dtype)


def shape(x):
    return tuple(x.shape)


def sin(x, name=None):
    return torch.sin(x)


def stack(values, axis=0, name=None):
    return torch.stack([_t(v, float32) for v in values], dim=axis)


def unstack(value, axis=0, name=None):
    return list(torch.unbind(value, dim=axis))


def transpose(a, perm=None, name=None):
    assert perm is None and a.dim() == 2
    return a.t()


def matmul(a, b, a_is_sparse=False, b_is_sparse=False, name=None):
    return torch.matmul(a, b)


def add_n(inputs, name=None):
    out = inputs[0]
    for x in inputs[1:]:
        out = out + x
    return out


def reduce_sum(x, axis=None, name=None):
    return torch.sum(x) if axis is None else torch.sum(x, dim=axis)


def multiply(a, b, name=None):
    return a * b


def add(a, b, name=None):
    return a + b


def subtract(a, b, name=None):
    return a - b


def square(x, name=None):
    return x * x


def concat(values, axis, name=None):
    return torch.cat(list(values), dim=axis)


def tile(x, multiples, name=None):
    return x.repeat(*[int(m) for m in multiples])


def reshape(x, shape, name=None):
    return torch.reshape(x, tuple(int(s) for s in shape))


def cast(x, dtype, name=None):
    return x.to(dtype)


def complex(real, imag, name=None):                 # noqa: A001 (TensorFlow's name)
    return torch.complex(real, imag)


def fft(x, name=None):
    return torch.fft.fft(x, dim=-1)


def complex_abs(x, name=None):
    return torch.abs(x)


@contextlib.contextmanager
def name_scope(name):
    yield


class _Placeholder(object):
    def __init__(self):
        self.value = None


def placeholder(dtype, shape=None, name=None):
    return _Placeholder()


class Graph(object):
    @contextlib.contextmanager
    def as_default(self):
        yield self


def _l2_loss(t, name=None):
    return torch.sum(t * t) / 2


nn = types.SimpleNamespace(l2_loss=_l2_loss)


class _ApplyOp(object):
    """What `opt.apply_gradients(grads_and_vars)` returns here: run(lr) performs ONE TF1 Adam update with the given gradients."""

    def __init__(self, opt, grads_and_vars):
        self.opt, self.gv = opt, grads_and_vars

    def run(self, lr):
        o = self.opt
        o.t += 1
        lr_t = lr * np.sqrt(1 - o.beta2 ** o.t) / (1 - o.beta1 ** o.t)
        with torch.no_grad():
            for i, (g, v) in enumerate(self.gv):
                if i not in o.m:
                    o.m[i], o.v[i] = torch.zeros_like(v), torch.zeros_like(v)
                o.m[i] = o.beta1 * o.m[i] + (1 - o.beta1) * g
                o.v[i] = o.beta2 * o.v[i] + (1 - o.beta2) * g * g
                v -= lr_t * o.m[i] / (torch.sqrt(o.v[i]) + o.epsilon)


class _AdamOptimizer(object):
    def __init__(self, learning_rate=0.001, beta1=0.9, beta2=0.999, epsilon=1e-8):
        self.learning_rate, self.beta1, self.beta2, self.epsilon = learning_rate, beta1, beta2, epsilon
        self.t, self.m, self.v = 0, {}, {}

    def compute_gradients(self, loss):
        tv = [v for v, trainable in _VARIABLES if trainable]
        grads = torch.autograd.grad(loss, tv, retain_graph=True, allow_unused=True)
        return [(g.detach(), v) for g, v in zip(grads, tv) if g is not None]

    def apply_gradients(self, grads_and_vars):
        return _ApplyOp(self, grads_and_vars)


class _Saver(object):
    pass


train = types.SimpleNamespace(AdamOptimizer=_AdamOptimizer, Saver=_Saver)


# ---- tensorflow.python.framework.function.Defun ---------------------------------------------------------------------------------------
class _DefunCall(object):
    """`@function.Defun(*dtypes, grad_func=g)`: the decorated Python function f becomes an op whose gradient is g(*inputs, grad) -- NOT the
    derivative of f (that is the whole point of the reference's matexp_op_grad / matvecexp_op_grad)."""

    def __init__(self, fn, grad_func):
        self.fn, self.grad_func = fn, grad_func

    def __call__(self, *inputs):
        fn, grad_func = self.fn, self.grad_func
        if grad_func is None:
            return fn(*inputs)

        class Op(torch.autograd.Function):
            @staticmethod
            def forward(ctx, *xs):
                ctx.save_for_backward(*xs)
                with torch.no_grad():
                    return fn(*xs)

            @staticmethod
            def backward(ctx, grad):
                xs = ctx.saved_tensors
                with torch.no_grad():
                    gs = grad_func(*xs, grad)
                return tuple(g if need else None for g, need in zip(gs, ctx.needs_input_grad))

        return Op.apply(*inputs)


def Defun(*dtypes, **kw):
    grad_func = kw.get('grad_func')

    def deco(fn):
        return _DefunCall(fn, grad_func)
    return deco


def reset():
    del _VARIABLES[:]
