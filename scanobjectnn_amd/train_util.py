"""Training-step semantics of the reference trainers, restated:
learning-rate / BN-decay schedules (`pointnet2/train.py:116-134`), TF-flavoured Adam
(`tf.train.AdamOptimizer` defaults, epsilon OUTSIDE the bias-corrected sqrt), on flat parameter /
gradient buffers so that one data-parallel step is: backward -> ONE all-reduce of a flat fp32 bucket
-> ONE fused optimiser update.
"""
import math
import os

import numpy as np
import torch

BASE_LEARNING_RATE = 1e-3   # train.py:31
DECAY_STEP = 200000         # train.py:34
DECAY_RATE = 0.7            # train.py:35
BN_INIT_DECAY = 0.5         # train.py:78
BN_DECAY_DECAY_RATE = 0.5   # train.py:79
BN_DECAY_CLIP = 0.99        # train.py:81
FUSED_ADAM = os.environ.get("PCOPS_FUSED_ADAM", "1") != "0"   # device buffers: the update as one libpcops launch


def get_learning_rate(global_step, batch_size, base_lr=BASE_LEARNING_RATE, decay_step=DECAY_STEP,
                      decay_rate=DECAY_RATE):
    """train.py:116-124: max(base * rate^floor(step*B/decay_step), 1e-5) (staircase)"""
    lr = base_lr * decay_rate ** math.floor(global_step * batch_size / decay_step)
    return max(lr, 0.00001)


def get_bn_decay(global_step, batch_size, bn_decay_decay_step=float(DECAY_STEP)):
    """train.py:126-134: min(0.99, 1 - 0.5 * 0.5^floor(step*B/decay_step))"""
    bn_momentum = BN_INIT_DECAY * BN_DECAY_DECAY_RATE ** math.floor(global_step * batch_size / bn_decay_decay_step)
    return min(BN_DECAY_CLIP, 1 - bn_momentum)


class FlatParams:
    """Re-homes every parameter of `module` (and its gradient) as a view into ONE flat fp32 buffer."""

    ALIGN = 4   # floats: every parameter starts on a 16-byte boundary (the kernels read weights 16 bytes at a time;
    #             a weight matrix at an odd offset silently takes the slow generic GEMM -- measured on DGCNN's
    #             320 -> 1024 layer: 3.4 ms instead of 2.3 ms).  The gaps stay zero in both buffers.

    def __init__(self, module):
        self.params = [p for p in module.parameters() if p.requires_grad]
        pad = lambda k: (k + self.ALIGN - 1) // self.ALIGN * self.ALIGN   # noqa: E731
        n = sum(pad(p.numel()) for p in self.params)
        dev = self.params[0].device
        self.flat = torch.zeros(n, dtype=torch.float32, device=dev)
        self.grad = torch.zeros(n, dtype=torch.float32, device=dev)
        self._zeros = torch.zeros(max(pad(p.numel()) for p in self.params), dtype=torch.float32, device=dev)
        self._gviews = []
        self._pads = []
        self._armed = False
        off = 0
        for p in self.params:
            k = p.numel()
            self.flat[off:off + k].copy_(p.data.reshape(-1))
            p.data = self.flat[off:off + k].view(p.shape)
            p.grad = self.grad[off:off + k].view(p.shape)
            self._gviews.append(p.grad)
            self._pads.append(pad(k) - k)
            off += pad(k)
        self.numel = n
        self._offs = []
        off = 0
        for p in self.params:
            self._offs.append(off)
            off += pad(p.numel())
        self._buckets = None             # overlapped all-reduce (enable_overlap)

    # -- gradient all-reduce overlapped with the backward pass ------------------------------------------------------
    def enable_overlap(self, world, nbuckets=4, force=False):
        """Cut the flat bucket into `nbuckets` contiguous ranges of about equal size (at parameter boundaries).  A range
        is packed and its all-reduce issued ASYNCHRONOUSLY as soon as autograd has produced the gradient of its last
        parameter -- the ranges that hold the late layers (produced first by the backward pass) travel while the
        kernels of the early layers still run; `collect_mean()` issues what is left, waits and scales.  xGMI is
        point-to-point and the whole bucket is 6-11 MB, so few, large ranges (latency-bound collectives) rather than
        DDP's 25 MB default or one collective per layer.  The launch order is the order autograd finishes the ranges:
        identical on every rank (same graph).  No-op for a single rank (force: take the path anyway -- the RCCL smoke
        test)."""
        import torch.distributed as dist
        if (world <= 1 and not force) or not dist.is_initialized():
            return self
        if getattr(self, "_hooks_registered", False):
            return self                  # hooks are permanent: a second registration would count every gradient twice
        target = max(1, self.numel // max(1, nbuckets))
        self._buckets = []               # [first param, one past last param, start offset, end offset]
        first = 0
        for i in range(len(self.params)):
            end = self._offs[i + 1] if i + 1 < len(self.params) else self.numel
            if end - self._offs[first] >= target or i + 1 == len(self.params):
                self._buckets.append((first, i + 1, self._offs[first], end))
                first = i + 1
        self._bucket_of = [0] * len(self.params)
        for b, (i0, i1, _, _) in enumerate(self._buckets):
            for i in range(i0, i1):
                self._bucket_of[i] = b
        self._world = world
        self._pending = [0] * len(self._buckets)
        self._works = []
        for i, p in enumerate(self.params):
            p.register_post_accumulate_grad_hook(lambda _p, i=i: self._grad_ready(i))
        self._hooks_registered = True
        return self

    def _grad_ready(self, i):
        if self._buckets is None or not self._armed:
            return
        b = self._bucket_of[i]
        self._pending[b] -= 1
        if self._pending[b] == 0:
            self._launch(b)

    def _pack(self, i0, i1, out):
        pieces = []
        for p, gap in zip(self.params[i0:i1], self._pads[i0:i1]):
            pieces.append(p.grad.reshape(-1) if p.grad is not None else self._zeros[:p.numel()])
            if gap:
                pieces.append(self._zeros[:gap])
        torch.cat(pieces, out=out)

    def _launch(self, b):
        import torch.distributed as dist
        i0, i1, s, e = self._buckets[b]
        self._pack(i0, i1, self.grad[s:e])
        work = dist.all_reduce(self.grad[s:e], op=dist.ReduceOp.SUM, async_op=True)
        self._works.append(work)
        from . import dist as D
        D.pending_work.append(work)      # a SyncBN exchange on a separate communicator orders itself behind these
        self._pending[b] = -1            # launched

    def collect_mean(self, world):
        """after backward(): the flat gradient averaged over the ranks -- `collect()` + one all-reduce, or, with
        `enable_overlap`, the ranges already under way plus whatever autograd did not reach."""
        from . import dist as D
        if self._buckets is None:
            return D.allreduce_mean_(self.collect(), world)
        self._armed = False
        for b in range(len(self._buckets)):
            if self._pending[b] >= 0:    # a parameter of this range got no gradient: zeros for it
                self._launch(b)
        for w in self._works:
            w.wait()
        del self._works[:]
        del D.pending_work[:]
        self.grad.mul_(1.0 / self._world)
        for p, v in zip(self.params, self._gviews):
            p.grad = v
        return self.grad

    def zero_grad(self):
        self.grad.zero_()
        for p, v in zip(self.params, self._gviews):
            p.grad = v

    def begin_step(self):
        """Drop the gradient views: autograd then hands each parameter its freshly computed gradient instead of
        adding it into a zeroed buffer (one fill + one add launch per parameter saved); `collect()` re-homes them."""
        for p in self.params:
            p.grad = None
        if self._buckets is not None:
            for b, (i0, i1, _, _) in enumerate(self._buckets):
                self._pending[b] = i1 - i0
            self._armed = True

    def collect(self):
        """after backward(): pack the per-parameter gradients into the flat bucket with ONE concatenation and make
        p.grad a view of it again (parameters the loss did not reach contribute zeros)"""
        pieces = []
        for p, gap in zip(self.params, self._pads):
            pieces.append(p.grad.reshape(-1) if p.grad is not None else self._zeros[:p.numel()])
            if gap:
                pieces.append(self._zeros[:gap])
        torch.cat(pieces, out=self.grad)
        for p, v in zip(self.params, self._gviews):
            p.grad = v
        return self.grad


class TFAdam:
    """tf.train.AdamOptimizer(lr): m,v EMA; lr_t = lr*sqrt(1-b2^t)/(1-b1^t); p -= lr_t*m/(sqrt(v)+eps)."""

    def __init__(self, flat, beta1=0.9, beta2=0.999, epsilon=1e-8):
        self.fp = flat
        self.b1, self.b2, self.eps = beta1, beta2, epsilon
        self.m = torch.zeros_like(flat.flat)
        self.v = torch.zeros_like(flat.flat)
        self.t = 0

    @torch.no_grad()
    def step(self, lr):
        if self.fp.flat.is_cuda and FUSED_ADAM:
            return self._step_device(lr)
        self.t += 1
        g = self.fp.grad
        # (1 - beta) in fp32, as TensorFlow's ApplyAdam forms it from its float32 hyper-parameter tensors: 1 - 0.999f is
        # 0.99998713e-3, not 1e-3 -- the device kernel does the same
        c1 = float(np.float32(1.0) - np.float32(self.b1))
        c2 = float(np.float32(1.0) - np.float32(self.b2))
        self.m.mul_(self.b1).add_(g, alpha=c1)
        self.v.mul_(self.b2).addcmul_(g, g, value=c2)
        lr_t = lr * math.sqrt(1 - self.b2 ** self.t) / (1 - self.b1 ** self.t)
        self.fp.flat.addcdiv_(self.m, self.v.sqrt().add_(self.eps), value=-lr_t)

    @torch.no_grad()
    def _step_device(self, lr):
        """the same update as ONE libpcops launch over the flat buffers (pcops_adam_step) instead of seven elementwise ones"""
        from . import _lib
        self.t += 1
        lr_t = lr * math.sqrt(1 - self.b2 ** self.t) / (1 - self.b1 ** self.t)
        _lib.call("pcops_adam_step", self.fp.numel, self.fp.flat.data_ptr(), self.fp.grad.data_ptr(), self.m.data_ptr(),
                  self.v.data_ptr(), float(self.b1), float(self.b2), float(lr_t), float(self.eps))


class TFMomentum:
    """tf.train.MomentumOptimizer(lr, momentum) (`pointnet2/train.py:165-166`): accum <- momentum*accum + g;
    p <- p - lr*accum (no Nesterov, no dampening) on the flat buffers."""

    def __init__(self, flat, momentum=0.9):
        self.fp = flat
        self.momentum = momentum
        self.accum = torch.zeros_like(flat.flat)

    @torch.no_grad()
    def step(self, lr):
        self.accum.mul_(self.momentum).add_(self.fp.grad)
        self.fp.flat.add_(self.accum, alpha=-lr)


def make_optimizer(name, flat, momentum=0.9):
    """`--optimizer adam|momentum` of the reference trainers (`pointnet2/train.py:41-42,165-168`)"""
    if name == "adam":
        return TFAdam(flat)
    if name == "momentum":
        return TFMomentum(flat, momentum)
    raise ValueError("optimizer must be 'adam' or 'momentum', got %r" % (name,))
