"""Checkpoint interchange by TF variable name.

The reference saves ``tf.train.Saver`` checkpoints (sqair/scripts/experiment.py:165-168, restored by
sqair/experiment_tools.py:56-144); their variables are the 2 951 522 parameters listed in
notebooks/play.ipynb:239-362.  ``sqair_amd.params.param_spec`` carries that TF name for every entry of the flat
parameter buffer, so a checkpoint here is an ``.npz`` keyed by TF variable name (plus optimiser slots and the global
step): anything that can enumerate a TF checkpoint — the five-line dump script in INTEGRATION.md — produces a file this
module loads, and files written here can be assigned back variable by variable.  TF itself is not needed (and not
available) on this side.
"""
from __future__ import annotations

import numpy as np

from .params import flatten_params, param_spec, unflatten_params


def tf_name_map(F, img_hw):
    """{our name: TF variable name} in flat-buffer order."""
    return {name: tf_name for name, shape, init, tf_name in param_spec(F, img_hw)}


def tf_shape(name, shape):
    """Shape of the TF variable behind a flat-buffer entry: identical, except that the decoder's mean image carries TF's
    channel axis ([H, W, 1], notebooks/play.ipynb:243; configs/mlp_mnist_model.py:77-80 adds it)."""
    return tuple(shape) + (1,) if name == "dec.mean_img" else tuple(shape)


def to_tf_dict(params, F, img_hw):
    """{TF variable name: array in the TF variable's shape} from a name->array dict or a flat vector."""
    spec = param_spec(F, img_hw)
    if not isinstance(params, dict):
        params = unflatten_params(np.asarray(params, dtype=np.float32), spec)
    out = {}
    for name, shape, init, tf_name in spec:
        if tf_name in out:
            raise ValueError("duplicate TF variable name {}".format(tf_name))
        out[tf_name] = np.asarray(params[name], dtype=np.float32).reshape(tf_shape(name, shape))
    return out


def from_tf_dict(tf_vars, F, img_hw, strict=True):
    """Inverse of ``to_tf_dict``.  Every array must have exactly the TF variable's shape (a transposed matrix with the
    right element count is an error, not a reshape); only the mean image may drop its trailing channel axis.  ``strict``
    additionally requires every variable to be present."""
    spec = param_spec(F, img_hw)
    out, missing = {}, []
    for name, shape, init, tf_name in spec:
        if tf_name not in tf_vars:
            missing.append(tf_name)
            continue
        v = np.asarray(tf_vars[tf_name], dtype=np.float32)
        if tuple(v.shape) != tf_shape(name, shape) and tuple(v.shape) != tuple(shape):
            raise ValueError("variable {} has shape {}, expected {}".format(tf_name, tuple(v.shape), tf_shape(name, shape)))
        out[name] = v.reshape(shape)
    if missing and strict:
        raise KeyError("checkpoint lacks {} variables, e.g. {}".format(len(missing), missing[:3]))
    return out


_SLOTS = {"rmsprop": ("RMSProp", "RMSProp_1"), "adam": ("Adam_1", "Adam"), "momentum": (None, "Momentum"), "sgd": (None, None)}


def save_checkpoint(path, core, optimizer=None, global_step=None, trainer=None):
    """Parameters of a ``SqairCore`` + (optionally) the optimiser slots of ``sqair_amd.train.Optimizer`` under TF's slot
    naming for the optimiser kind (RMSProp: ``<var>/RMSProp`` = mean square, ``<var>/RMSProp_1`` = momentum; Adam:
    ``<var>/Adam`` = m, ``<var>/Adam_1`` = v plus its step count as ``beta1_power`` / ``beta2_power``; Momentum:
    ``<var>/Momentum``) + ``global_step``.  With ``trainer`` the optimiser and the step counter (it drives the LR
    schedule, the curriculum and the Philox noise key) are taken from it."""
    if trainer is not None:
        optimizer = trainer.opt if optimizer is None else optimizer
        global_step = trainer.step_no if global_step is None else global_step
    flat = core.flat.detach().cpu().numpy()
    blob = to_tf_dict(flat, core.F, (core.H, core.W))
    if optimizer is not None:
        spec = core.spec
        ms_slot, mom_slot = _SLOTS[optimizer.kind]
        ms = unflatten_params(optimizer.ms.detach().cpu().numpy(), spec)
        mom = unflatten_params(optimizer.mom.detach().cpu().numpy(), spec)
        for name, shape, init, tf_name in spec:
            if ms_slot:
                blob[tf_name + "/" + ms_slot] = ms[name].reshape(tf_shape(name, shape))
            if mom_slot:
                blob[tf_name + "/" + mom_slot] = mom[name].reshape(tf_shape(name, shape))
        if optimizer.kind == "adam":
            blob["beta1_power"] = np.asarray(0.9 ** optimizer.t, dtype=np.float64)
            blob["beta2_power"] = np.asarray(0.999 ** optimizer.t, dtype=np.float64)
        blob["optimizer_step"] = np.asarray(int(optimizer.t), dtype=np.int64)
    blob["global_step"] = np.asarray(int(global_step or 0), dtype=np.int64)
    np.savez(path, **blob)


def load_checkpoint(path, core, optimizer=None, strict=True, trainer=None):
    """Loads parameters (and slots, when present and an optimiser is given) into the core; returns global_step.  With
    ``trainer`` its optimiser slots, Adam step count and ``step_no`` are restored as well."""
    import torch
    if trainer is not None and optimizer is None:
        optimizer = trainer.opt
    with np.load(path) as z:
        tf_vars = {k: z[k] for k in z.files}
    params = from_tf_dict(tf_vars, core.F, (core.H, core.W), strict=strict)
    merged = core.get_params()
    merged.update(params)
    core.set_params(merged)
    step = int(tf_vars.get("global_step", 0))
    if optimizer is not None:
        spec = core.spec
        ms_slot, mom_slot = _SLOTS[optimizer.kind]
        for slot, buf in ((ms_slot, optimizer.ms), (mom_slot, optimizer.mom)):
            if slot and all((tf_name + "/" + slot) in tf_vars for _, _, _, tf_name in spec):
                vals = {name: np.asarray(tf_vars[tf_name + "/" + slot]).reshape(shape) for name, shape, _, tf_name in spec}
                buf.copy_(torch.from_numpy(flatten_params(vals, spec)))
        # number of updates applied so far: drives Adam's bias correction and, for every kind, the default learning-rate
        # schedule of apply_gradients(lr=None).  Our own files store it; a TF dump has global_step (one update per step in the
        # reference's loop, experiment.py:150-155) and, for Adam, beta1_power = 0.9^t — which TF keeps in fp32 and which
        # underflows to 0 after ~1000 steps, so it is only used while it is still a usable number.
        if "optimizer_step" in tf_vars:
            optimizer.t = int(tf_vars["optimizer_step"])
        else:
            optimizer.t = step
            b1p = float(tf_vars["beta1_power"]) if (optimizer.kind == "adam" and "beta1_power" in tf_vars) else 0.0
            if step == 0 and np.isfinite(b1p) and 1e-30 < b1p < 1.0:
                optimizer.t = int(round(np.log(b1p) / np.log(0.9)))
    if trainer is not None:
        trainer.step_no = step
    return step
