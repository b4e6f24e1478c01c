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


def to_tf_dict(params, F, img_hw):
    """{TF variable name: array in the TF variable's shape} from a name->array dict or a flat vector."""
    spec = param_spec(F, img_hw)
    if not isinstance(params, dict):
        params = unflatten_params(np.asarray(params, dtype=np.float32), spec)
    out = {}
    for name, shape, init, tf_name in spec:
        if tf_name in out:
            raise ValueError("duplicate TF variable name {}".format(tf_name))
        out[tf_name] = np.asarray(params[name], dtype=np.float32).reshape(shape)
    return out


def from_tf_dict(tf_vars, F, img_hw, strict=True):
    """Inverse of ``to_tf_dict``; ``strict`` requires every variable to be present with the right number of elements."""
    spec = param_spec(F, img_hw)
    out, missing = {}, []
    for name, shape, init, tf_name in spec:
        if tf_name not in tf_vars:
            missing.append(tf_name)
            continue
        v = np.asarray(tf_vars[tf_name], dtype=np.float32)
        if v.size != int(np.prod(shape)) if len(shape) else v.size != 1:
            raise ValueError("variable {} has {} elements, expected shape {}".format(tf_name, v.size, shape))
        out[name] = v.reshape(shape)
    if missing and strict:
        raise KeyError("checkpoint lacks {} variables, e.g. {}".format(len(missing), missing[:3]))
    return out


def save_checkpoint(path, core, optimizer=None, global_step=0):
    """Parameters of a ``SqairCore`` (and, optionally, the optimiser slots of ``sqair_amd.train.Optimizer`` under the
    TF slot naming ``<var>/RMSProp`` = mean square, ``<var>/RMSProp_1`` = momentum) + ``global_step``."""
    flat = core.flat.detach().cpu().numpy()
    blob = to_tf_dict(flat, core.F, (core.H, core.W))
    if optimizer is not None:
        spec = core.spec
        ms = unflatten_params(optimizer.ms.detach().cpu().numpy(), spec)
        mom = unflatten_params(optimizer.mom.detach().cpu().numpy(), spec)
        for name, shape, init, tf_name in spec:
            blob[tf_name + "/RMSProp"] = ms[name]
            blob[tf_name + "/RMSProp_1"] = mom[name]
    blob["global_step"] = np.asarray(int(global_step), dtype=np.int64)
    np.savez(path, **blob)


def load_checkpoint(path, core, optimizer=None, strict=True):
    """Loads parameters (and slots, when present and an optimiser is given) into the core; returns global_step."""
    import torch
    with np.load(path) as z:
        tf_vars = {k: z[k] for k in z.files}
    params = from_tf_dict(tf_vars, core.F, (core.H, core.W), strict=strict)
    merged = core.get_params()
    merged.update(params)
    core.set_params(merged)
    if optimizer is not None:
        spec = core.spec
        have = all((tf_name + "/RMSProp") in tf_vars for _, _, _, tf_name in spec)
        if have:
            ms = {name: tf_vars[tf_name + "/RMSProp"] for name, _, _, tf_name in spec}
            mom = {name: tf_vars[tf_name + "/RMSProp_1"] for name, _, _, tf_name in spec}
            optimizer.ms.copy_(torch.from_numpy(flatten_params(ms, spec)))
            optimizer.mom.copy_(torch.from_numpy(flatten_params(mom, spec)))
    return int(tf_vars.get("global_step", 0))
