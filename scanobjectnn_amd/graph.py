"""TF1-style variable store on torch, so that the reference's functional layer API
(`tf_util.conv2d(inputs, ..., scope=...)`, `pointnet_sa_module(..., scope='layer1')`,
`get_model(point_cloud, is_training, bn_decay)`) can be kept call-for-call.

Variables are created on first use under the '/'-joined scope name the reference would
give them (`layer1/conv0/weights`, `layer1/conv0/bn/beta`, `fc1/biases`, ...: the names a
TF checkpoint of the reference holds) and reused on every later call, i.e.
the graph is define-by-run with persistent variables.
"""
import contextlib
import math
import threading

import torch

_tls = threading.local()


class Graph(torch.nn.Module):
    def __init__(self, device=None, seed=None):
        super().__init__()
        self._device = torch.device(device) if device is not None else None
        self._scope = []
        self._gen = None
        if seed is not None:
            self._gen = torch.Generator(device="cpu")
            self._gen.manual_seed(int(seed))
        self.end_points = {}

    # -- scoping -----------------------------------------------------------------
    @contextlib.contextmanager
    def as_default(self):
        prev = getattr(_tls, "graph", None)
        _tls.graph = self
        saved = list(self._scope)
        try:
            yield self
        finally:
            self._scope = saved
            _tls.graph = prev

    def full_name(self, name):
        return "/".join(self._scope + [name])

    # -- variables ---------------------------------------------------------------
    def _materialise(self, shape, init):
        t = torch.empty(tuple(shape), dtype=torch.float32)
        init(t, self._gen)
        return t.to(self._device) if self._device is not None else t

    def get_variable(self, name, shape, initializer, trainable=True):
        full = self.full_name(name)
        if trainable:
            if full in self._parameters:
                p = self._parameters[full]
            else:
                p = torch.nn.Parameter(self._materialise(shape, initializer))
                self.register_parameter(full, p)
        else:
            if full in self._buffers:
                p = self._buffers[full]
            else:
                p = self._materialise(shape, initializer)
                self.register_buffer(full, p)
        if tuple(p.shape) != tuple(shape):
            raise ValueError("variable %s exists with shape %s, requested %s"
                             % (full, tuple(p.shape), tuple(shape)))
        return p


def get_default_graph():
    g = getattr(_tls, "graph", None)
    if g is None:
        raise RuntimeError("no default Graph: call layers inside `with graph.as_default():` "
                           "(scanobjectnn_amd.graph.Model does this for get_model functions)")
    return g


@contextlib.contextmanager
def variable_scope(name):
    g = get_default_graph()
    g._scope.append(name)
    try:
        yield g.full_name("")[:-1]
    finally:
        g._scope.pop()


def get_variable(name, shape, initializer, trainable=True):
    return get_default_graph().get_variable(name, shape, initializer, trainable)


# -- initialisers (callables (tensor, generator) -> None) ------------------------------
def constant_initializer(value):
    def init(t, gen):
        t.fill_(float(value))
    return init


def xavier_initializer():
    """tf.contrib.layers.xavier_initializer(uniform=True): U(-l, l), l = sqrt(6/(fan_in+fan_out)),
    fans computed TF-style: receptive field x in/out depth for [kh,kw,cin,cout] kernels."""
    def init(t, gen):
        shape = t.shape
        rf = 1
        for s in shape[:-2]:
            rf *= s
        fan_in, fan_out = shape[-2] * rf, shape[-1] * rf
        limit = math.sqrt(6.0 / (fan_in + fan_out))
        t.uniform_(-limit, limit, generator=gen)
    return init


def truncated_normal_initializer(stddev):
    def init(t, gen):
        torch.nn.init.trunc_normal_(t, mean=0.0, std=stddev, a=-2 * stddev, b=2 * stddev, generator=gen)
    return init


class Model(torch.nn.Module):
    """Wraps a reference-style `get_model(point_cloud, is_training, bn_decay=None, **kw)` into a
    torch module that owns the Graph.  Variables appear at the first call: run `build(example)`
    before constructing an optimiser."""

    def __init__(self, get_model, device=None, seed=None, **kwargs):
        super().__init__()
        self.graph = Graph(device=device, seed=seed)
        self._get_model = get_model
        self._kwargs = kwargs

    def forward(self, point_cloud, is_training, bn_decay=None):
        with self.graph.as_default():
            return self._get_model(point_cloud, is_training, bn_decay=bn_decay, **self._kwargs)

    @torch.no_grad()
    def build(self, example):
        self.forward(example, is_training=False)
        return self
