"""Train-step semantics of the reference driver, host side.

reference: sqair/scripts/experiment.py:126-155 — global step, piecewise-constant learning rate
(``schedule`` = comma-separated relative segment lengths; the rate is divided by 3 at every boundary),
``tf.train.RMSPropOptimizer(lr, momentum=.9)``; sqair/data/mnist_tools.py:84-92 — sequence-length
curriculum (``seq_len`` grows by one every ``stage_itr`` iterations).  The parameter update itself is the
fused HIP kernel ``sqair_rmsprop_step``; the backward pass that feeds it is built up in sqair_bwd.hip
(decoder branch done, recurrent part next round)."""
from __future__ import annotations

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
    i = int(np.searchsorted(np.asarray(bounds), step, side="right"))
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
