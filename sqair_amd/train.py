"""Train-step semantics of the reference driver, host side.

reference: sqair/scripts/experiment.py:126-155 — global step, piecewise-constant learning rate
(``schedule`` = comma-separated relative segment lengths; the rate is divided by 3 at every boundary),
``tf.train.RMSPropOptimizer(lr, momentum=.9)``; sqair/data/mnist_tools.py:84-92 — sequence-length
curriculum (``seq_len`` grows by one every ``stage_itr`` iterations).  The parameter update itself is the
fused HIP kernel ``sqair_rmsprop_step``; the gradients come from ``sqair_backward`` (csrc/sqair_train.hip).
``Optimizer`` mirrors the ``opt.compute_gradients`` / ``opt.apply_gradients`` pair the reference's ``make_target``
and driver use; ``Trainer`` is the driver's inner loop (experiment.py:150-185) for one rank."""
from __future__ import annotations

import ctypes as C

import numpy as np


def lr_schedule(F):
    """Returns (boundaries, values) of the piecewise-constant learning rate (experiment.py:127-138):
    schedule '4,6,10' over train_itr iterations -> boundaries at cumulative 4/20 and 10/20 of train_itr,
    values lr, lr/3, lr/9."""
    sched = [float(s) for s in str(F.schedule).split(",")]
    total = sum(sched)
    bounds = [int(round(v)) for v in (np.cumsum(sched)[:-1] / total * int(F.train_itr))]
    values = [float(F.learning_rate) * (1.0 / 3.0) ** i for i in range(len(sched))]
    return bounds, values


def learning_rate(F, step):
    bounds, values = lr_schedule(F)
    # tf.train.piecewise_constant: values[0] for x <= boundaries[0], values[i] for boundaries[i-1] < x <= boundaries[i]
    i = int(np.searchsorted(np.asarray(bounds), step, side="left"))
    return values[i]


def curriculum_seq_len(F, step, max_len):
    """Sequence-length curriculum (mnist_tools.py:84-92): seq_len + step // stage_itr, capped at the data length;
    seq_len = 0 or stage_itr = 0 disables it."""
    if int(F.seq_len) <= 0 or int(F.stage_itr) <= 0:
        return int(max_len)
    return int(min(max_len, int(F.seq_len) + step // int(F.stage_itr)))


def rmsprop_reference(theta, grad, ms, mom, lr, decay=0.9, momentum=0.9, eps=1e-10):
    """NumPy restatement of the TF update used by the tests."""
    ms = decay * ms + (1.0 - decay) * grad * grad
    mom = momentum * mom + lr * grad / np.sqrt(ms + eps)
    return theta - mom, ms, mom


class Optimizer(object):
    """The optimisers the reference driver offers (experiment.py:139-147) on the flat fp32 buffers of a ``SqairCore``.
    ``rmsprop`` (the default and the only one the shipped configs use) is the fused HIP kernel; ``adam`` / ``sgd`` /
    ``momentum`` are a handful of elementwise torch ops on the same device buffers (TF1 defaults)."""

    def __init__(self, core, kind="rmsprop", momentum=0.9, decay=0.9, epsilon=1e-10):
        import torch
        self.core, self.kind = core, str(kind).lower()
        if self.kind not in ("rmsprop", "adam", "sgd", "momentum"):
            raise ValueError("unknown optimiser {!r}".format(kind))
        self.momentum, self.decay, self.epsilon = float(momentum), float(decay), float(epsilon)
        # TF initialises the RMSProp mean-square slot with ones, Adam's second moment with zeros
        self.ms = torch.ones_like(core.flat) if self.kind == "rmsprop" else torch.zeros_like(core.flat)
        self.mom = torch.zeros_like(core.flat)
        self.t = 0

    def compute_gradients(self, model, l2_reg=0.0):
        """``opt.compute_gradients(target)`` of the reference (model.py:159-161): evaluates the VIMCO target and the gradient
        of every trainable variable for the model's current batch; returns the (gradient, variable name) list."""
        return model.make_target(self, l2_reg=l2_reg)[1]

    def apply_gradients(self, grads, lr=None, grad_scale=1.0, global_step=None):
        """theta <- theta - update(grad_scale * grad); re-packs the weights for the next forward pass.  ``grads`` is the
        core's flat gradient tensor or the (gradient, name) list ``Model.make_target`` / ``compute_gradients`` return (its
        entries are views of that flat tensor).  ``lr`` defaults to the piecewise schedule of the core's flags at
        ``global_step`` (experiment.py:127-138; default: the number of updates applied so far)."""
        import torch
        from . import _capi
        core = self.core
        if isinstance(grads, (list, tuple)):
            if len(grads) != len(core.spec):
                raise ValueError("expected one (gradient, name) pair per variable ({}), got {}".format(len(core.spec), len(grads)))
            flat_grad = core.flat_grad
        else:
            flat_grad = grads
        if lr is None:
            lr = learning_rate(core.F, self.t if global_step is None else int(global_step))
        self.t += 1
        if self.kind == "rmsprop":
            with torch.cuda.device(core.device):
                core.check(core.lib.sqair_rmsprop_step(
                    core.handle, core.flat.data_ptr(), flat_grad.data_ptr(), self.ms.data_ptr(), self.mom.data_ptr(),
                    core.n_params, float(lr), self.decay, self.momentum, self.epsilon, float(grad_scale),
                    C.c_void_p(torch.cuda.current_stream(core.device).cuda_stream)), "sqair_rmsprop_step")
        else:
            g = flat_grad * float(grad_scale) if grad_scale != 1.0 else flat_grad
            if self.kind == "sgd":
                core.flat.add_(g, alpha=-float(lr))
            elif self.kind == "momentum":      # tf.train.MomentumOptimizer: accum = m accum + g; theta -= lr accum
                self.mom.mul_(self.momentum).add_(g)
                core.flat.add_(self.mom, alpha=-float(lr))
            else:                              # tf.train.AdamOptimizer defaults beta1 .9, beta2 .999, eps 1e-8
                b1, b2, eps = 0.9, 0.999, 1e-8
                self.mom.mul_(b1).add_(g, alpha=1.0 - b1)
                self.ms.mul_(b2).addcmul_(g, g, value=1.0 - b2)
                lr_t = float(lr) * np.sqrt(1.0 - b2 ** self.t) / (1.0 - b1 ** self.t)
                core.flat.addcdiv_(self.mom, self.ms.sqrt().add_(eps), value=-lr_t)
        core.pack()


class Trainer(object):
    """One rank of the reference's training loop (experiment.py:150-185): per step draw noise, evaluate the VIMCO
    target and its gradients on this rank's shard of sequences (ONE HIP graph replay), all-reduce the flat gradient
    buffer over the ranks (the single collective of the step, RCCL over xGMI), apply the optimiser.
    ``model`` is a ``sqair_amd.model.Model`` bound to this rank's shard."""

    def __init__(self, model, F, use_graph=True, comm=None, collective=True):
        """comm: a ``sqair_amd.rccl.RcclComm`` — the gradient all-reduce is then enqueued on the core's own stream
        (``ncclAllReduce``); None = the default ``torch.distributed`` group, if one is initialised.  ``collective=False``: a
        trainer that belongs to ONE rank of a running job (bench.py's single-GPU reference and timeline legs) and must not
        enter the job's collective."""
        self.model, self.core, self.F = model, model.core, F
        self.comm = comm
        self.collective = bool(collective)
        self.opt = Optimizer(self.core, getattr(F, "opt", "rmsprop"))
        self.step_no = 0
        self.use_graph = bool(use_graph)

    def step(self, obs=None, noise=None, generator=None, seed=None, global_batch=None, b0=0, presence=None):
        """One training step, asynchronous on the core's stream (``core.stream.synchronize()`` or read metrics inside
        ``core.on_stream()`` to observe results).  Returns the flat gradient buffer: on a multi-rank job it holds the SUM over
        the ranks of the shard gradients (the all-reduce's result); the 1 / world that makes it the reference's ``reduce_mean``
        over the global batch is applied inside the optimiser kernel (``grad_scale``), so a caller that logs or clips these
        values must scale them by ``1 / world`` itself."""
        import torch
        from . import _capi
        from .dist import allreduce_flat_grads
        core, F = self.core, self.F
        if obs is not None:
            obs = torch.as_tensor(obs, dtype=torch.float32)
            if obs.dim() == 5:
                obs = obs[..., 0]
            if int(obs.shape[0]) != core.T or int(obs.shape[1]) != core.B:
                # sequence-length curriculum (mnist_tools.py:80-92) or a new batch size: re-bind through the Model so that
                # its n_timesteps / batch_size / obs / ground truth follow (the gradient graph is re-captured on the next
                # evaluation)
                self.model.rebind(obs, presence=presence)
                obs = None
            else:
                self.model.obs = obs.to(core.device)
                if presence is not None:
                    self.model.gt_presence = torch.as_tensor(presence, dtype=torch.float32).to(core.device)
        with core.on_stream():
            if obs is not None:
                core.obs.copy_(self.model.obs.reshape(core.obs.shape))
            if noise is not None:
                core.noise.copy_(torch.as_tensor(noise, dtype=torch.float32).reshape(core.noise.shape))
            elif seed is not None:  # library Philox keyed by (seed, step, position in the global batch)
                core.draw_noise(seed=seed, step=self.step_no, global_batch=global_batch, b0=b0)
            else:
                core.draw_noise(generator)
            g = core.grad_step(use_graph=self.use_graph)
            l2 = float(getattr(F, "l2", 0.0))
            if l2 != 0.0:
                core.check(core.lib.sqair_add_l2_grad(
                    core.handle, core.flat.data_ptr(), g.data_ptr(), core.n_params, l2, core._stream()), "sqair_add_l2_grad")
            scale = allreduce_flat_grads(g, comm=self.comm, stream=core.stream) if self.collective else 1.0
            self.opt.apply_gradients(g, learning_rate(F, self.step_no), grad_scale=scale)
        self.step_no += 1
        return g
