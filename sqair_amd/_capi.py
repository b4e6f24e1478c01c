"""ctypes binding of libsqair_hip.so (C-ABI declared in include/sqair_hip.h).

The library is built in-tree by ``__graft_entry__.build()`` / ``sqair_amd/csrc/build.py``.  There is
no CPU fallback: if the shared object is missing the import of the HIP path fails loudly.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libsqair_hip.so")

OUTPUT_FIELDS = [
    "what", "what_loc", "what_scale", "where", "where_loc", "where_scale", "presence_prob", "presence",
    "presence_logit", "obj_id", "step_log_prob", "canvas", "glimpse", "disc_what_log_prob",
    "disc_where_log_prob", "disc_what_prior_log_prob", "disc_where_prior_log_prob", "disc_log_prob",
    "disc_prior_log_prob", "disc_prob", "prop_what_log_prob", "prop_where_log_prob",
    "prop_what_prior_log_prob", "prop_where_prior_log_prob", "prop_log_prob", "prop_prior_log_prob",
    "prop_prob", "discrete_log_prob", "num_prop_steps_per_sample", "num_disc_steps_per_sample",
    "num_steps_per_sample", "prop_pres", "disc_pres", "data_ll_per_sample", "kl_per_sample",
    "log_q_z_given_x_per_sample", "log_p_z_per_sample", "log_weights_per_timestep",
    "final_temporal_state", "final_prior_state", "final_last_used_id",
]
N_REFERENCE_OUTPUTS = 38  # the TensorArrays of reference sqair/seq.py:121-177


class SqairConfig(C.Structure):
    _fields_ = [
        ("img_h", C.c_int32), ("img_w", C.c_int32), ("glimpse_size", C.c_int32),
        ("n_steps_per_image", C.c_int32), ("n_what", C.c_int32), ("n_hidden", C.c_int32),
        ("k_particles", C.c_int32), ("prop_prior_type", C.c_int32), ("disc_prior_type", C.c_int32),
        ("masked_glimpse", C.c_int32), ("rec_where_prior", C.c_int32),
        ("prop_prior_step_bias", C.c_float), ("step_success_prob", C.c_float), ("output_std", C.c_float),
        ("background_std", C.c_float), ("where_prior_mean", C.c_float * 4),
        ("sample_from_prior", C.c_int32), ("generate_after", C.c_int32), ("time_cell", C.c_int32), ("prior_cell", C.c_int32), ("rnn_cell", C.c_int32),
    ]


class SqairOutputs(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in OUTPUT_FIELDS]


_PROTOS = {
    "sqair_abi_version": (C.c_int, []),
    "sqair_build_id": (C.c_char_p, []),
    "sqair_build_flags": (C.c_char_p, []),
    "sqair_create": (C.c_int, [C.POINTER(SqairConfig), C.POINTER(C.c_void_p)]),
    "sqair_destroy": (C.c_int, [C.c_void_p]),
    "sqair_last_error": (C.c_char_p, [C.c_void_p]),
    "sqair_param_count": (C.c_int64, [C.c_void_p]),
    "sqair_param_entries": (C.c_int, [C.c_void_p]),
    "sqair_param_entry": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_char_p), C.POINTER(C.c_int64),
                                    C.POINTER(C.c_int64)]),
    "sqair_packed_bytes": (C.c_int64, [C.c_void_p]),
    "sqair_workspace_bytes": (C.c_int64, [C.c_void_p, C.c_int, C.c_int]),
    "sqair_noise_width": (C.c_int, [C.c_void_p]),
    "sqair_pack_params": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "sqair_forward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int,
                                C.c_int, C.POINTER(SqairOutputs), C.c_void_p, C.c_int64, C.c_void_p]),
    "sqair_train_workspace_bytes": (C.c_int64, [C.c_void_p, C.c_int, C.c_int]),
    "sqair_forward_train": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int,
                                      C.c_int, C.POINTER(SqairOutputs), C.c_void_p, C.c_int64, C.c_void_p]),
    "sqair_graph_capture": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                                      C.c_int, C.c_int, C.POINTER(SqairOutputs), C.c_void_p, C.c_int64,
                                      C.c_void_p]),
    "sqair_graph_launch": (C.c_int, [C.c_void_p, C.c_void_p]),
    "sqair_graph_nodes": (C.c_int, [C.c_void_p]),
    "sqair_elbo": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                             C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_void_p), C.c_int, C.c_void_p,
                             C.c_void_p]),
    "sqair_st_crop": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    "sqair_st_insert_loglik": (C.c_int, [C.c_void_p] * 8 + [C.c_int, C.c_void_p]),
    "sqair_linear_test": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int,
                                    C.c_int, C.c_int, C.c_void_p, C.c_int64, C.c_void_p]),
    "sqair_gru_test": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int,
                                 C.c_void_p, C.c_int64, C.c_void_p]),
    "sqair_set_workspace_clearing": (C.c_int, [C.c_void_p, C.c_int]),
    "sqair_clear_workspace": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "sqair_set_option": (C.c_int, [C.c_void_p, C.c_char_p, C.c_int]),
    "sqair_chain_status": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "sqair_check_scales": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "sqair_check_finite": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_char_p, C.c_void_p, C.c_void_p]),
    "sqair_timeline_available": (C.c_int, []),
    "sqair_timeline_begin": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64]),
    "sqair_timeline_count": (C.c_int, [C.c_void_p]),
    "sqair_timeline_end": (C.c_int, [C.c_void_p]),
    "sqair_timeline_record": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_char_p), C.POINTER(C.c_int64), C.POINTER(C.c_int),
                                        C.POINTER(C.c_int)]),
    "sqair_lstm_test": (C.c_int, [C.c_void_p] * 6 + [C.c_int, C.c_int, C.c_void_p, C.c_int64, C.c_void_p]),
    "sqair_lstm_cell_bwd_test": (C.c_int, [C.c_void_p] * 7 + [C.c_int, C.c_void_p]),
    "sqair_get_config": (C.c_int, [C.c_void_p, C.POINTER(SqairConfig)]),
    "sqair_st_crop_bwd": (C.c_int, [C.c_void_p] * 7 + [C.c_int, C.c_void_p]),
    "sqair_st_insert_loglik_bwd": (C.c_int, [C.c_void_p] * 11 + [C.c_int64, C.c_int, C.c_void_p]),
    "sqair_elbo_bwd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                 C.c_void_p]),
    "sqair_backward_scratch_bytes": (C.c_int64, [C.c_void_p, C.c_int, C.c_int]),
    "sqair_backward_decoder": (C.c_int, [C.c_void_p] * 6 + [C.c_int, C.c_int, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64,
                                         C.c_void_p, C.c_void_p, C.c_void_p]),
    "sqair_backward_bytes": (C.c_int64, [C.c_void_p, C.c_int, C.c_int]),
    "sqair_backward": (C.c_int, [C.c_void_p] * 7 + [C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64,
                                 C.c_void_p, C.c_void_p]),
    "sqair_set_generation_noise": (C.c_int, [C.c_void_p, C.c_void_p]),
    "sqair_fill_noise": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_uint64, C.c_uint64, C.c_void_p]),
    "sqair_capture_begin": (C.c_int, [C.c_void_p, C.c_void_p]),
    "sqair_capture_end": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int]),
    "sqair_capture_launch": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p]),
    "sqair_add_l2_grad": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_float, C.c_void_p]),
    "sqair_rmsprop_step": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_float,
                                     C.c_float, C.c_float, C.c_float, C.c_float, C.c_void_p]),
    "sqair_linear_bwd_test": (C.c_int, [C.c_void_p] * 8 + [C.c_int] * 4 + [C.c_void_p, C.c_int64, C.c_void_p]),
    "sqair_debug_linear_time": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int,
                                          C.c_void_p, C.c_int64, C.c_int, C.POINTER(C.c_float), C.c_void_p]),
    "sqair_debug_linear_graph_time": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int,
                                              C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_float), C.c_void_p]),
    "sqair_debug_dense_log": (C.c_int, [C.c_void_p, C.c_int]),
    "sqair_debug_dense_log_entry": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_int)]),
    "sqair_debug_layers": (C.c_int, [C.c_void_p]),
    "sqair_debug_padded_count": (C.c_int64, [C.c_void_p, C.POINTER(C.c_int)]),
    "sqair_debug_layer": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int),
                                    C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "sqair_debug_plan": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int),
                                   C.POINTER(C.c_int)]),
}

EXPORTED_SYMBOLS = [n for n in _PROTOS if not n.startswith("sqair_debug")]

TIMELINE_LIB_PATH = os.path.join(_HERE, "libsqair_hip_timeline.so")
# the same sources compiled with -DSQAIR_WIDE: the rest of the reference's flag range (n_what up to 128, up to 16 object slots,
# n_units up to 16) on a larger slot record and plain-loop per-row kernels; same C-ABI, slower
WIDE_LIB_PATH = os.path.join(_HERE, "libsqair_hip_wide.so")
# what the product library is laid out for (csrc/sqair_glue.h: SQ_MAXN, SQ_MAX_NWHAT, SQ_MAX_NHIDDEN)
PRODUCT_LIMITS = dict(n_what=50, n_steps_per_image=8, n_hidden=256)
WIDE_LIMITS = dict(n_what=128, n_steps_per_image=14, n_hidden=512)   # (15, 16 slots: the log-probability adjoint's LDS staging does not fit the wide record)


def lib_path_for(n_what, n_steps_per_image, n_hidden):
    """The library a configuration runs on: the product library inside its limits, the wide build beyond them."""
    inside = (n_what <= PRODUCT_LIMITS["n_what"] and n_steps_per_image <= PRODUCT_LIMITS["n_steps_per_image"] and
              n_hidden <= PRODUCT_LIMITS["n_hidden"])
    return LIB_PATH if inside else WIDE_LIB_PATH

ABI_VERSION = 2
_VARIANT_OF = {"libsqair_hip.so": "product", "libsqair_hip_timeline.so": "timeline", "libsqair_hip_knobs.so": "knobs",
               "libsqair_hip_wide.so": "wide"}

_libs = {}


class StaleLibraryError(ImportError):
    """The shared object was not compiled from the sources beside it."""


def source_id():
    """Hash of the kernel sources ON DISK (sqair_amd/csrc/ + include/sqair_hip.h + compiler flags): what a fresh build of any
    variant would report as its `sqair_build_id()`."""
    from .csrc import build as _b
    if not any(f.endswith(".hip") for f in os.listdir(_b.HERE)):
        raise OSError("no kernel sources in {}".format(_b.HERE))   # (build.py alone does not make a source tree)
    return _b.source_id()


def lib(path=None, allow_stale=False):
    """Loads the shared library (once per path).  Raises ImportError with the build hint if it is absent, if its ABI version is
    not this binding's, or (StaleLibraryError) if the build id compiled into it is not the hash of the sources on disk -- a
    binary is git-ignored and travels beside the sources, so nothing else ties the two together.  `path` selects a build
    VARIANT of the same sources (TIMELINE_LIB_PATH: every wave stamps its start / end, measurement only); the default is the
    product library.  There is no environment override."""
    path = path or LIB_PATH
    if path not in _libs:
        if not os.path.exists(path):
            raise ImportError(
                "{} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` or "
                "`python sqair_amd/csrc/build.py [--timeline]`; sqair_amd has no CPU fallback".format(path))
        # torch first: its wheel bundles its own HIP runtime; loading ours before it would put two runtimes into the
        # process (the second one then reports "no ROCm-capable device")
        import torch  # noqa: F401
        l = C.CDLL(path)
        # the version first, through the one symbol every version has: a binding that has drifted fails HERE, not in a getattr
        l.sqair_abi_version.restype = C.c_int
        if l.sqair_abi_version() != ABI_VERSION:
            raise ImportError("{}: ABI version {} but this binding speaks {}; rebuild (python sqair_amd/csrc/build.py)".format(
                os.path.basename(path), l.sqair_abi_version(), ABI_VERSION))
        for name, (res, args) in _PROTOS.items():
            fn = getattr(l, name)
            fn.restype = res
            fn.argtypes = args
        got = l.sqair_build_id().decode()
        try:
            want = source_id()
        except (OSError, ImportError):   # a binary-only deployment (no csrc/ -- ModuleNotFoundError --, its .hip files stripped, or no
            # include/ beside the package): nothing to compare with
            want = None
            if not allow_stale:
                import warnings
                warnings.warn("{}: the kernel sources are not beside the package, the staleness check of the binary "
                              "(build id {}) is skipped".format(os.path.basename(path), got))
        if want is not None and got != want and not allow_stale:
            raise StaleLibraryError(
                "{} was compiled from sources {} but the sources on disk are {}: rebuild (python sqair_amd/csrc/build.py "
                "--force [--timeline] [--knobs]); numbers measured on a stale binary would be attributed to the wrong "
                "code".format(os.path.basename(path), got, want))
        variant = _VARIANT_OF.get(os.path.basename(path))
        if variant is not None and l.sqair_build_flags().decode() != variant:
            raise ImportError("{} reports build variant '{}', expected '{}'".format(
                os.path.basename(path), l.sqair_build_flags().decode(), variant))
        _libs[path] = l
    return _libs[path]


def build_id(path=None):
    """The build id COMPILED INTO the loaded library (`sqair_build_id()`: the hash of the sources that binary was built from).
    Profiles carry it so that a number is only ever quoted next to the binary it was measured on; `lib()` has already checked
    that it equals `source_id()`."""
    return lib(path).sqair_build_id().decode()


def check(handle, rc, what, library=None):
    """Raises RuntimeError with the handle's error text.  `library` must be the shared object the handle came from (a handle of
    the timeline / knob variant is not the product library's to read): SqairCore.check passes its own."""
    if rc != 0:
        msg = (library or lib()).sqair_last_error(handle)
        raise RuntimeError("{} failed (rc={}): {}".format(what, rc, msg.decode() if msg else ""))
