"""Host-side mirror of the reference's operator API for the SQAIR hot path.

``load(img, coords, num, mean_img=None, debug=False)`` and the ``Model`` it returns keep the call
surface of reference sqair/configs/mlp_mnist_model.py:74-150 and sqair/model.py:33-214 (attribute
names, shapes, the ``make_target`` / ``resample`` methods, flag names) while every quantity is
computed by libsqair_hip.so on the MI355X.  PyTorch is plumbing only: device memory, the current
HIP stream, the noise draw.  There is no CPU fallback; without the library or a GPU the
constructor raises.

Differences a TF1 user will notice (TF builds a graph, ``sess.run`` evaluates it): here
``model.run()`` plays the role of ``sess.run`` — it executes the forward pass once and refreshes
every attribute (``model.elbo_iwae`` ...).  The first attribute access runs it implicitly.
"""
from __future__ import annotations

import ctypes as C
import math
import os

import numpy as np
import torch

from . import _capi
from .flags import FLAGS, get_params, parse_string_flag
from .params import flatten_params, init_params, param_offsets, param_spec, unflatten_params

_PRIOR_TYPES = {"rnn": 0, "rw": 1, "guided": 2}
_DISC_PRIOR_TYPES = {"cat": 0, "geom": 1}


def make_config(F, img_hw):
    """Flags -> SqairConfig.  Raises the reference's errors for invalid choices
    (reference: sqair/propagate.py:42-43, sqair/sqair_modules.py:224)."""
    if F.prop_prior_type not in _PRIOR_TYPES:
        raise ValueError('Invalid prior type: "{}". Choose from {}.'.format(F.prop_prior_type, list(_PRIOR_TYPES)))
    if F.disc_prior_type not in _DISC_PRIOR_TYPES:
        raise ValueError("Invalid prior type: {}".format(F.disc_prior_type))
    cells = ("VanillaRNN", "GRU", "LSTM")
    if F.transition not in cells or F.time_transition not in cells or F.prior_transition not in cells:
        raise NotImplementedError(
            "HIP path implements transition, time_transition and prior_transition in {{VanillaRNN, GRU, LSTM}} "
            "(configs/mlp_mnist_model.py:86-87,125 pick Sonnet cells by name); got {}/{}/{}".format(
                F.transition, F.time_transition, F.prior_transition))
    p = get_params(F)
    if int(F.n_units) < 1:
        raise ValueError("n_units must be >= 1; got {}".format(F.n_units))
    sp = parse_string_flag(F.scale_prior, num_elements=2)
    std = float(np.float32(np.float32(np.sqrt(F.output_std)) ** np.float32(2.0)))  # modules.py:419-422
    return _capi.SqairConfig(
        int(img_hw[0]), int(img_hw[1]), int(F.glimpse_size), int(F.n_steps_per_image), int(F.n_what),
        int(p.n_hidden), int(F.k_particles), _PRIOR_TYPES[F.prop_prior_type], _DISC_PRIOR_TYPES[F.disc_prior_type],
        int(bool(F.masked_glimpse)), int(bool(F.rec_where_prior)), float(F.prop_prior_step_bias),
        float(F.step_success_prob), std, std, (C.c_float * 4)(sp[0], sp[1], 0.0, 0.0),
        int(bool(F.sample_from_prior)), int(getattr(F, "generate_after", -1)), {"GRU": 0, "LSTM": 1, "VanillaRNN": 2}[F.time_transition],
        {"GRU": 0, "LSTM": 1, "VanillaRNN": 2}[F.prior_transition], {"VanillaRNN": 0, "LSTM": 1, "GRU": 2}[F.transition])


class SqairCore(object):
    """Thin owner of a library handle + the device buffers it needs (parameters, packed parameters,
    workspace, noise, outputs) for one (T, B) shape on one device / stream."""

    def __init__(self, F, img_hw, device="cuda:0", lib_path=None, stream_priority=0, options=None):
        if not torch.cuda.is_available():
            raise RuntimeError("sqair_amd needs a HIP device (no CPU fallback)")
        self.F = F
        self.cfg = make_config(F, img_hw)
        # lib_path: a build variant (sqair_amd.timeline).  Default: the product library for everything it is laid out for, the
        # wide build of the same sources for the rest of the flag range (n_what > 50, more than 8 slots, n_units > 8)
        self.lib = _capi.lib(lib_path or _capi.lib_path_for(self.cfg.n_what, self.cfg.n_steps_per_image, self.cfg.n_hidden))
        self.device = torch.device(device)
        self.handle = C.c_void_p()
        rc = self.lib.sqair_create(C.byref(self.cfg), C.byref(self.handle))
        if rc != 0:
            raise ValueError("sqair_create rejected the configuration (rc={}): limits n_what <= {n_what}, n_steps_per_image <= "
                             "{n_steps_per_image}, n_units <= 16, k_particles <= 256".format(
                                 rc, **_capi.WIDE_LIMITS))
        # documented run-time options of the library (include/sqair_hip.h: sqair_set_option), set before any workspace is sized
        self.options = dict(options or {})
        for name, value in self.options.items():
            self.check(self.lib.sqair_set_option(self.handle, name.encode(), int(value)), "sqair_set_option")
        self.spec = param_spec(F, img_hw)
        self.offsets, self.n_params = param_offsets(self.spec)
        assert self.n_params == self.lib.sqair_param_count(self.handle), "parameter inventory mismatch"
        self.N = int(F.n_steps_per_image)
        self.K = int(F.k_particles)
        self.nw = int(F.n_what)
        self.nh = get_params(F).n_hidden
        self.snh = self.nh * (2 if F.time_transition == "LSTM" else 1)  # temporal state: [hidden | cell] for an LSTM
        self.psnh = self.nh * (2 if F.prior_transition == "LSTM" else 1)  # same for the propagation prior's state
        self.G = int(F.glimpse_size)
        self.H, self.W = int(img_hw[0]), int(img_hw[1])
        self.nzw = self.lib.sqair_noise_width(self.handle)
        with torch.cuda.device(self.device):
            self.flat = torch.zeros(self.n_params, dtype=torch.float32, device=self.device)
            self.packed = torch.zeros(self.lib.sqair_packed_bytes(self.handle) // 4, dtype=torch.float32,
                                      device=self.device)
            # all launches go to one dedicated non-default stream (HIP refuses to capture the legacy stream)
            self.stream = torch.cuda.Stream(device=self.device, priority=int(stream_priority))
        self._shape = None
        self._graph_ready = False
        self.check(self.lib.sqair_set_workspace_clearing(self.handle, 0), "sqair_set_workspace_clearing")

    def _clear_ws(self, ws, train):
        """One workspace per (T, B, inference | training) is cleared ONCE here; the per-pass zero fill of the library is
        switched off for this handle (include/sqair_hip.h: sqair_set_workspace_clearing)."""
        self.check(self.lib.sqair_clear_workspace(self.handle, ws.data_ptr(), ws.numel() * 4, self.T_bind, self.B_bind,
                                                                 int(train), self._stream()), "sqair_clear_workspace")
        self.stream.synchronize()

    def __del__(self):
        try:
            if self.handle:
                self.lib.sqair_destroy(self.handle)
                self.handle = C.c_void_p()
        except Exception:
            pass

    def _stream(self):
        return C.c_void_p(self.stream.cuda_stream)

    def on_stream(self):
        """Context manager making the core's stream torch's current stream.  Run whole loops (noise draw, forward /
        gradient evaluation, optimiser, metric reads) inside it: with a second active stream (e.g. the default one,
        joined by events every step) every kernel of the replayed graphs measures ~1 us slower on this stack
        (forward 6.0 -> 5.0 ms, training step 15.6 -> 13.6 ms at cfg-2)."""
        return torch.cuda.stream(self.stream)

    def _join_in(self):
        cur = torch.cuda.current_stream(self.device)
        if cur != self.stream:
            self.stream.wait_stream(cur)

    def _join_out(self):
        cur = torch.cuda.current_stream(self.device)
        if cur != self.stream:
            cur.wait_stream(self.stream)

    # ---- parameters ------------------------------------------------------------------------------
    def set_params(self, params):
        """params: dict name -> array (sqair_amd.params naming) or a flat float32 vector."""
        if isinstance(params, dict):
            flat = flatten_params(params, self.spec)
        else:
            flat = np.asarray(params, dtype=np.float32)
        assert flat.shape == (self.n_params,)
        self.flat.copy_(torch.from_numpy(flat))
        self.pack()

    def get_params(self):
        return unflatten_params(self.flat.cpu().numpy(), self.spec)

    def pack(self):
        with torch.cuda.device(self.device):
            self._join_in()
            self.check(self.lib.sqair_pack_params(self.handle, self.flat.data_ptr(),
                                                                self.packed.data_ptr(), self._stream()),
                        "sqair_pack_params")
            self._join_out()
        self._graph_ready = False

    # ---- buffers for a (T, B) shape ----------------------------------------------------------------
    def bind(self, T, B, outputs="all"):
        if self._shape == (T, B, outputs if isinstance(outputs, str) else tuple(outputs)):
            return
        N, K, nw, G, H, W, nh = self.N, self.K, self.nw, self.G, self.H, self.W, self.nh
        R = B * K
        self.T_bind, self.B_bind = int(T), int(B)
        shapes = dict(
            what=(T, R, N, nw), what_loc=(T, R, N, nw), what_scale=(T, R, N, nw), where=(T, R, N, 4),
            where_loc=(T, R, N, 4), where_scale=(T, R, N, 4), presence_prob=(T, R, N), presence=(T, R, N),
            presence_logit=(T, R, N), obj_id=(T, R, N), step_log_prob=(T, R), canvas=(T, R, H, W),
            glimpse=(T, R, N, G, G), disc_what_log_prob=(T, R, N), disc_where_log_prob=(T, R, N),
            disc_what_prior_log_prob=(T, R, N), disc_where_prior_log_prob=(T, R, N), disc_log_prob=(T, R),
            disc_prior_log_prob=(T, R), disc_prob=(T, R, N + 1), prop_what_log_prob=(T, R, N),
            prop_where_log_prob=(T, R, N), prop_what_prior_log_prob=(T, R, N), prop_where_prior_log_prob=(T, R, N),
            prop_log_prob=(T, R), prop_prior_log_prob=(T, R), prop_prob=(T, R, N), discrete_log_prob=(T, R),
            num_prop_steps_per_sample=(T, R), num_disc_steps_per_sample=(T, R), num_steps_per_sample=(T, R),
            prop_pres=(T, R, N), disc_pres=(T, R, N), data_ll_per_sample=(T, R), kl_per_sample=(T, R),
            log_q_z_given_x_per_sample=(T, R), log_p_z_per_sample=(T, R), log_weights_per_timestep=(T, R),
            final_temporal_state=(R, N, self.snh), final_prior_state=(R, N, self.psnh), final_last_used_id=(R,),
        )
        if outputs == "all":
            wanted = list(_capi.OUTPUT_FIELDS)
        elif outputs == "minimal":  # what the ELBO, the VIMCO target and the scalar metrics need
            wanted = ["log_weights_per_timestep", "discrete_log_prob", "data_ll_per_sample", "kl_per_sample",
                      "log_q_z_given_x_per_sample", "log_p_z_per_sample", "num_steps_per_sample",
                      "num_disc_steps_per_sample", "num_prop_steps_per_sample"]
        else:
            wanted = list(outputs)
        with torch.cuda.device(self.device):
            self.out = {n: torch.zeros(shapes[n], dtype=torch.float32, device=self.device) for n in wanted}
            self.obs = torch.zeros(T, B, H, W, dtype=torch.float32, device=self.device)
            self.noise = torch.zeros(T, R, 2, N, self.nzw, dtype=torch.float32, device=self.device)
            self.gen_noise = None
            if self.cfg.sample_from_prior:  # second set of draws: the prior samples of the generation modes
                self.gen_noise = torch.zeros(T, R, 2, N, self.nzw, dtype=torch.float32, device=self.device)
                self.check(self.lib.sqair_set_generation_noise(self.handle, self.gen_noise.data_ptr()),
                            "sqair_set_generation_noise")
            self.ws_bytes = self.lib.sqair_workspace_bytes(self.handle, T, B)
            self.workspace = torch.empty(self.ws_bytes // 4, dtype=torch.float32, device=self.device)
            self._clear_ws(self.workspace, False)
            # ELBO outputs
            self.log_weights = torch.zeros(B, K, dtype=torch.float32, device=self.device)
            self.elbo_iwae_per_example = torch.zeros(B, dtype=torch.float32, device=self.device)
            self.importance_weights = torch.zeros(B, K, dtype=torch.float32, device=self.device)
            self.vimco_signal = torch.zeros(B, K, dtype=torch.float32, device=self.device)
            self.scalars = torch.zeros(16, dtype=torch.float32, device=self.device)
            self.iw_means = torch.zeros(8, dtype=torch.float32, device=self.device)
        self.c_out = _capi.SqairOutputs(**{n: (self.out[n].data_ptr() if n in self.out else None)
                                          for n in _capi.OUTPUT_FIELDS})
        self.mean_names = [n for n in ("data_ll_per_sample", "log_p_z_per_sample", "log_q_z_given_x_per_sample",
                                       "kl_per_sample", "num_steps_per_sample", "num_disc_steps_per_sample",
                                       "num_prop_steps_per_sample") if n in self.out]
        self.c_means = (C.c_void_p * 8)(*([self.out[n].data_ptr() for n in self.mean_names] +
                                         [None] * (8 - len(self.mean_names))))
        self.T, self.B = T, B
        self._shape = (T, B, outputs if isinstance(outputs, str) else tuple(outputs))
        self._graph_ready = False
        self._train_graph_ready = False
        self.train_ws = None
        self.bwd_scratch = None

    def draw_noise(self, generator=None, seed=None, step=0, global_batch=None, b0=0):
        """eps ~ N(0,1) for the Normals, u ~ U[0,1) for the presence Bernoullis, on device.  With ``seed`` the library's own
        Philox generator fills the buffer in ONE launch, keyed by (seed, step, position in the global batch) so that every
        data-parallel rank draws the rows one GPU would have drawn; otherwise torch's generator is used."""
        if seed is None:
            self.noise.normal_(generator=generator)
            self.noise[..., -1].uniform_(generator=generator)
            if self.gen_noise is not None:
                self.gen_noise.normal_(generator=generator)
                self.gen_noise[..., -1].uniform_(generator=generator)
            return
        with torch.cuda.device(self.device):
            self._join_in()
            self.check(self.lib.sqair_fill_noise(
                self.handle, self.noise.data_ptr(), self.T, self.B, int(global_batch or self.B), int(b0), int(seed), int(step),
                self._stream()), "sqair_fill_noise")
            if self.gen_noise is not None:  # the prior samples of the generation modes: an independent Philox key
                self.check(self.lib.sqair_fill_noise(
                    self.handle, self.gen_noise.data_ptr(), self.T, self.B, int(global_batch or self.B), int(b0),
                    int(seed) ^ 0x9E3779B97F4A7C15, int(step), self._stream()), "sqair_fill_noise")
            self._join_out()

    # ---- execution ---------------------------------------------------------------------------------
    def _args(self, t_offset):
        return (self.handle, self.flat.data_ptr(), self.packed.data_ptr(), self.obs.data_ptr(),
                self.noise.data_ptr(), self.T, self.B, int(t_offset), C.byref(self.c_out),
                self.workspace.data_ptr(), self.ws_bytes, self._stream())

    def forward(self, t_offset=0, use_graph=False, train=False):
        """Launches the whole T-frame forward pass + the ELBO reductions on the current stream.  ``train`` keeps the
        tape for the backward pass (larger workspace, allocated on first use)."""
        with torch.cuda.device(self.device):
            self._join_in()
            if train:
                nb = self.lib.sqair_train_workspace_bytes(self.handle, self.T, self.B)
                if getattr(self, "train_ws", None) is None or self.train_ws.numel() * 4 < nb:
                    self.train_ws = torch.empty(nb // 4, dtype=torch.float32, device=self.device)
                    self._clear_ws(self.train_ws, True)
                args = list(self._args(t_offset))
                args[9], args[10] = self.train_ws.data_ptr(), nb
                self.check(self.lib.sqair_forward_train(*args), "sqair_forward_train")
            elif use_graph:
                if not self._graph_ready:
                    torch.cuda.synchronize(self.device)
                    self.check(self.lib.sqair_graph_capture(*self._args(t_offset)), "sqair_graph_capture")
                    self._graph_ready = True
                self.check(self.lib.sqair_graph_launch(self.handle, self._stream()), "sqair_graph_launch")
            else:
                self.check(self.lib.sqair_forward(*self._args(t_offset)), "sqair_forward")
            dlp = self.out["discrete_log_prob"].data_ptr() if "discrete_log_prob" in self.out else None
            self.check(self.lib.sqair_elbo(
                self.handle, self.out["log_weights_per_timestep"].data_ptr(), dlp, self.T, self.B,
                self.log_weights.data_ptr(), self.elbo_iwae_per_example.data_ptr(),
                self.importance_weights.data_ptr(), self.vimco_signal.data_ptr(), self.scalars.data_ptr(),
                self.c_means, len(self.mean_names), self.iw_means.data_ptr(), self._stream()), "sqair_elbo")
            self._join_out()

    def backward(self, t_offset=0):
        """Full backward pass after forward(train=True): gradient of the VIMCO target / T w.r.t. every parameter.
        Leaves it in ``self.flat_grad`` (flat layout) on the core's stream; returns that tensor."""
        assert getattr(self, "train_ws", None) is not None, "backward() needs forward(train=True) first"
        with torch.cuda.device(self.device):
            nb = self.lib.sqair_backward_bytes(self.handle, self.T, self.B)
            if getattr(self, "bwd_scratch", None) is None or self.bwd_scratch.numel() * 4 < nb:
                self.bwd_scratch = torch.empty(nb // 4, dtype=torch.float32, device=self.device)
            if getattr(self, "flat_grad", None) is None:
                self.flat_grad = torch.zeros_like(self.flat)
            self._join_in()
            self.check(self.lib.sqair_backward(
                self.handle, self.flat.data_ptr(), self.packed.data_ptr(), self.obs.data_ptr(), self.noise.data_ptr(),
                self.importance_weights.data_ptr(), self.vimco_signal.data_ptr(), self.T, self.B, int(t_offset),
                self.train_ws.data_ptr(), self.train_ws.numel() * 4, self.bwd_scratch.data_ptr(), nb,
                self.flat_grad.data_ptr(), self._stream()), "sqair_backward")
            self._join_out()
        return self.flat_grad

    def grad_step(self, t_offset=0, use_graph=True):
        """One gradient evaluation = forward(train) + ELBO / VIMCO reductions + backward, replayed as ONE HIP graph
        (``use_graph``) once captured for the bound (T, B) shape.  The parameters must have been packed (set_params /
        pack) and obs / noise filled.  Returns the flat gradient buffer (valid on the current stream)."""
        if not use_graph:
            self.forward(t_offset=t_offset, train=True)
            return self.backward(t_offset=t_offset)
        if not getattr(self, "_train_graph_ready", False) or self._train_graph_key != (self._shape, int(t_offset)):
            # first call: run eagerly once (allocates tape / scratch, uploads the plan), then capture
            self.forward(t_offset=t_offset, train=True)
            self.backward(t_offset=t_offset)
            torch.cuda.synchronize(self.device)
            with torch.cuda.device(self.device):
                self.check(self.lib.sqair_capture_begin(self.handle, self._stream()), "sqair_capture_begin")
                try:
                    self._issue_train(t_offset)
                finally:
                    n = self.lib.sqair_capture_end(self.handle, self._stream(), 1)
                if n < 0:
                    self.check(n, "sqair_capture_end")
            self.train_graph_nodes = n
            self._train_graph_ready = True
            self._train_graph_key = (self._shape, int(t_offset))
        with torch.cuda.device(self.device):
            self._join_in()
            self.check(self.lib.sqair_capture_launch(self.handle, 1, self._stream()), "sqair_capture_launch")
            self._join_out()
        return self.flat_grad

    def _issue_train(self, t_offset):
        """The raw call sequence of a gradient evaluation on the core's stream (no stream joins: capturable)."""
        nb = self.train_ws.numel() * 4
        args = list(self._args(t_offset))
        args[9], args[10] = self.train_ws.data_ptr(), nb
        self.check(self.lib.sqair_forward_train(*args), "sqair_forward_train")
        dlp = self.out["discrete_log_prob"].data_ptr() if "discrete_log_prob" in self.out else None
        self.check(self.lib.sqair_elbo(
            self.handle, self.out["log_weights_per_timestep"].data_ptr(), dlp, self.T, self.B,
            self.log_weights.data_ptr(), self.elbo_iwae_per_example.data_ptr(),
            self.importance_weights.data_ptr(), self.vimco_signal.data_ptr(), self.scalars.data_ptr(),
            self.c_means, len(self.mean_names), self.iw_means.data_ptr(), self._stream()), "sqair_elbo")
        self.check(self.lib.sqair_backward(
            self.handle, self.flat.data_ptr(), self.packed.data_ptr(), self.obs.data_ptr(), self.noise.data_ptr(),
            self.importance_weights.data_ptr(), self.vimco_signal.data_ptr(), self.T, self.B, int(t_offset),
            self.train_ws.data_ptr(), nb, self.bwd_scratch.data_ptr(), self.bwd_scratch.numel() * 4,
            self.flat_grad.data_ptr(), self._stream()), "sqair_backward")

    def grads_by_name(self):
        """The last backward()'s gradients as a dict name -> tensor (reference variable shapes)."""
        out = {}
        for name, (o, shape) in self.offsets.items():
            n = int(np.prod(shape)) if len(shape) else 1
            out[name] = self.flat_grad[o:o + n].reshape(shape)
        return out

    def backward_decoder(self):
        """Decoder branch of the backward pass (first slice of the training step, see include/sqair_hip.h):
        gradients of the VIMCO target w.r.t. the decoder parameters as a dict name -> tensor, plus the seed
        gradients on the merged latents.  Requires a preceding forward() whose outputs did not request `glimpse`."""
        assert "glimpse" not in self.out, "bind(outputs=...) without 'glimpse' so that the glimpses stay in the workspace"
        R, M = self.B * self.K, self.B * self.K * self.N
        with torch.cuda.device(self.device):
            nb = self.lib.sqair_backward_scratch_bytes(self.handle, self.T, self.B)
            scratch = torch.empty(nb // 4, dtype=torch.float32, device=self.device)
            flat_grad = torch.zeros_like(self.flat)
            d_rec = torch.zeros(self.T, M, 64, dtype=torch.float32, device=self.device)
            self._join_in()
            self.check(self.lib.sqair_backward_decoder(
                self.handle, self.flat.data_ptr(), self.packed.data_ptr(), self.obs.data_ptr(),
                self.importance_weights.data_ptr(), self.vimco_signal.data_ptr(), self.T, self.B,
                self.workspace.data_ptr(), self.ws_bytes, scratch.data_ptr(), nb, flat_grad.data_ptr(), d_rec.data_ptr(),
                self._stream()), "sqair_backward_decoder")
            self._join_out()
            torch.cuda.synchronize(self.device)
        grads = {}
        for name, (o, shape) in self.offsets.items():
            if name.startswith("dec."):
                n = int(np.prod(shape)) if len(shape) else 1
                grads[name] = flat_grad[o:o + n].reshape(shape).clone()
        return grads, d_rec

    def graph_nodes(self):
        return self.lib.sqair_graph_nodes(self.handle)

    def set_vi_target(self, name):
        """The learning signal of the fused ELBO kernel: "vimco" (what the reference's make_target uses, targets.py:62-75) or
        "reinforce" (targets.py:78-89, the other entry of Model.VI_TARGETS).  Captured graphs are dropped: the signal is part of
        what they replay."""
        code = {"vimco": 0, "iwae": 0, "reinforce": 1}[name]
        self.check(self.lib.sqair_set_option(self.handle, b"vi_target", code), "sqair_set_option")
        self.vi_target = "reinforce" if code else "vimco"
        self._graph_ready = False
        self._train_graph_ready = False

    def check(self, rc, what):
        """Raises RuntimeError with this handle's error text, read through the library the handle came from."""
        _capi.check(self.handle, rc, what, library=self.lib)

    def check_scales(self, train=False):
        """Debug mode, the validate_args half: the posterior scales of `what` / `where` of every propagation and discovery slot of
        the last pass -- masked slots included -- are positive and finite (sqair/core.py:226, :261, sqair/modules.py:318-320)."""
        if getattr(self, "_finite_flag", None) is None:
            self._finite_flag = torch.zeros(2, dtype=torch.int32, device=self.device)
        ws = self.train_ws if train else self.workspace
        with torch.cuda.device(self.device):
            self._join_in()
            self.check(self.lib.sqair_check_scales(self.handle, ws.data_ptr(), self.T, self.B, int(train), self._finite_flag.data_ptr(),
                                                   self._stream()), "sqair_check_scales")

    def check_chain(self, train=False):
        """Raises when a launch of the in-launch slot chain (option slot_chain) did not complete in the last pass."""
        ws = self.train_ws if train else self.workspace
        if ws is None:
            return
        with torch.cuda.device(self.device):
            self.check(self.lib.sqair_chain_status(self.handle, ws.data_ptr(), self.T, self.B, int(train), self._stream()),
                       "sqair_chain_status")

    def check_finite(self, tensor, what):
        """Debug mode (reference: `debug` -> validate_args / allow_nan_stats=False, sqair/core.py:226, :261,
        sqair/modules.py:318-320): raises RuntimeError through the library's error channel when `tensor` holds NaN / Inf.
        Synchronises the core's stream."""
        if getattr(self, "_finite_flag", None) is None:
            self._finite_flag = torch.zeros(2, dtype=torch.int32, device=self.device)
        with torch.cuda.device(self.device):
            self._join_in()
            self.check(self.lib.sqair_check_finite(
                self.handle, tensor.data_ptr(), tensor.numel(), what.encode(), self._finite_flag.data_ptr(), self._stream()),
                "sqair_check_finite")


class Model(object):
    """Mirror of reference sqair/model.py:33-214 on top of ``SqairCore``."""
    VI_TARGETS = "iwae reinforce".split()
    TARGETS = VI_TARGETS

    def __init__(self, obs, coords, core, k_particles, presence=None, is_training=None, debug=False,
                 outputs="all"):
        self.core = core
        self.sequence = core
        self.k_particles = int(k_particles)
        assert self.k_particles == core.K
        obs = torch.as_tensor(obs, dtype=torch.float32)
        if obs.dim() == 5:  # [T,B,H,W,1]
            obs = obs[..., 0]
        self.obs = obs.to(core.device)
        self.coords = coords
        self.gt_presence = None if presence is None else torch.as_tensor(presence, dtype=torch.float32).to(core.device)
        self.debug = debug
        self.n_timesteps, self.batch_size = int(obs.shape[0]), int(obs.shape[1])
        self.img_size = list(obs.shape[2:])
        self.tiled_batch_size = self.batch_size * self.k_particles
        core.bind(self.n_timesteps, self.batch_size, outputs)
        core.obs.copy_(self.obs)
        self._ran = False
        self._use_graph = False

    def rebind(self, obs, presence=None):
        """New input tensors of a possibly different [T, B] (the sequence-length curriculum, mnist_tools.py:80-92, or a new
        batch size): the reference re-slices its input pipeline; here the core's buffers are re-bound and the model's
        own shape-dependent attributes follow, so that metrics normalise by the CURRENT T and compare against the
        CURRENT frames."""
        core = self.core
        obs = torch.as_tensor(obs, dtype=torch.float32)
        if obs.dim() == 5:
            obs = obs[..., 0]
        self.obs = obs.to(core.device)
        if presence is not None:
            self.gt_presence = torch.as_tensor(presence, dtype=torch.float32).to(core.device)
        elif self.gt_presence is not None and tuple(self.gt_presence.shape[:2]) != tuple(obs.shape[:2]):
            self.gt_presence = None   # ground truth of another batch: the accuracy is undefined until a new one is given
            for name in ("num_step_accuracy_per_example", "raw_num_step_accuracy", "num_step_accuracy"):
                self.__dict__.pop(name, None)  # (stale values of the previous shape)
        self.n_timesteps, self.batch_size = int(obs.shape[0]), int(obs.shape[1])
        self.tiled_batch_size = self.batch_size * self.k_particles
        # the previous step may still be running on the core's stream (Trainer.step is asynchronous): nothing is freed or
        # re-allocated before it has drained, and the new buffers are created and filled on that same stream
        core.stream.synchronize()
        with core.on_stream():
            core.bind(self.n_timesteps, self.batch_size, core._shape[2] if core._shape else "all")
            core.obs.copy_(self.obs)
        self._ran = False

    # `sess.run` --------------------------------------------------------------------------------------
    def run(self, noise=None, generator=None, resample_u=None, use_graph=None, gen_noise=None):
        """``sess.run`` of the whole output dict: synchronous like its reference counterpart (returns when the results
        are in the output tensors).  Everything is issued on the core's own stream."""
        core = self.core
        if use_graph is not None:
            self._use_graph = bool(use_graph)
        with core.on_stream():
            if noise is not None:
                core.noise.copy_(torch.as_tensor(noise, dtype=torch.float32).reshape(core.noise.shape))
                if gen_noise is not None:
                    core.gen_noise.copy_(torch.as_tensor(gen_noise, dtype=torch.float32).reshape(core.noise.shape))
            else:
                core.draw_noise(generator)
            core.forward(use_graph=self._use_graph)
            if self.debug:
                self._debug_checks()
            self._collect(resample_u)
        core.stream.synchronize()
        self._ran = True
        return self

    def _debug_checks(self, train=False):
        """`debug=True` of the reference turns on argument validation / NaN checks of every distribution inside the graph
        (sqair/core.py:226, :261, sqair/modules.py:318-320) — a failed check aborts `sess.run`.  Here: the per-frame
        log-weights (every log-probability of the pass ends up in them) and the sequence log-weights must be finite."""
        core = self.core
        # (train: the pass just run wrote the TRAINING workspace -- the tape --, not the inference one)
        core.check_chain(train=train)    # in-launch slot chain, when switched on: every launch of the pass completed
        core.check_scales(train=train)   # validate_args: every slot's what / where scale > 0 and finite -- the specific message first
        core.check_finite(core.out["log_weights_per_timestep"], "log_weights_per_timestep [T, B*K]")
        core.check_finite(core.log_weights, "log_weights [B, K]")

    def _collect(self, resample_u=None):
        core = self.core
        T, B, K = self.n_timesteps, self.batch_size, self.k_particles
        self.outputs = core.out
        for k, v in core.out.items():
            setattr(self, k, v)
        sc = core.scalars
        self.log_weights = core.log_weights
        self.elbo_vae = sc[0]
        self.elbo_iwae_per_example = core.elbo_iwae_per_example
        self.elbo_iwae = sc[1]
        self.normalised_elbo_vae = self.elbo_vae / float(T)
        self.normalised_elbo_iwae = self.elbo_iwae / float(T)
        self.importance_weights = core.importance_weights
        self.ess = sc[3]
        self.vimco_target = sc[2]
        self.vimco_signal = core.vimco_signal
        names = dict(data_ll_per_sample="data_ll", log_p_z_per_sample="log_p_z",
                     log_q_z_given_x_per_sample="log_q_z_given_x", kl_per_sample="kl",
                     num_steps_per_sample="num_steps", num_disc_steps_per_sample="num_disc_steps",
                     num_prop_steps_per_sample="num_prop_steps")
        for i, n in enumerate(core.mean_names):
            setattr(self, names[n], core.iw_means[i])
        # importance resampling index (reference draws it with tf Categorical; here inverse-CDF on u)
        if resample_u is None:
            u = torch.rand(B, 1, device=core.device)
        else:
            u = torch.as_tensor(resample_u, dtype=torch.float32, device=core.device).reshape(B, 1)
        cdf = torch.cumsum(self.importance_weights, -1)
        self.iw_resampling_idx = (cdf <= u).sum(-1).clamp(max=K - 1)
        if "canvas" in core.out:
            # model.py:112-115 against the K-tiled observation — broadcast over the particle axis, never materialised (a1)
            cv = core.out["canvas"].reshape(T, B, K, core.H, core.W)
            self.mse_per_sample = ((self.obs[:, :, None] - cv) ** 2).mean((0, 3, 4)).reshape(B * K)
            self.mse = self._imp_weighted_mean(self.mse_per_sample)
            self.raw_mse = self.mse_per_sample.mean()
        if self.gt_presence is not None and "num_steps_per_sample" in core.out:
            gt = self.gt_presence.sum(-1)
            ns = core.out["num_steps_per_sample"].reshape(-1, B, K)
            self.num_step_accuracy_per_example = (gt[..., None] == ns).float()
            self.raw_num_step_accuracy = self.num_step_accuracy_per_example.mean()
            self.num_step_accuracy = self._imp_weighted_mean(self.num_step_accuracy_per_example)
        for name in "obj_id canvas glimpse presence_prob presence presence_logit where".split():
            if name in core.out:
                setattr(self, "resampled_" + name, self.resample(core.out[name], axis=1))

    def __getattr__(self, name):
        # attribute access before the first run triggers it (graph-mode users expect attributes to exist)
        if name.startswith("_") or name in ("core", "outputs"):
            raise AttributeError(name)
        if not self.__dict__.get("_ran", False):
            self.run()
            return getattr(self, name)
        raise AttributeError(name)

    def _imp_weighted_mean(self, tensor):
        """reference: sqair/model.py:202-205."""
        B, K = self.batch_size, self.k_particles
        tensor = tensor.reshape(-1, B, K).mean(0)
        return (self.importance_weights * tensor * K).mean()

    def resample(self, *args, **kwargs):
        """reference: sqair/model.py:170-192 (gather along `axis` at b*K + iw_resampling_idx)."""
        axis = kwargs.pop("axis", -1)
        idx = self.iw_resampling_idx + torch.arange(self.batch_size, device=self.core.device) * self.k_particles
        res = [a.index_select(axis if axis >= 0 else a.dim() + axis, idx) if self.k_particles > 1 else a
               for a in args]
        return res[0] if len(res) == 1 else res

    def make_target(self, opt=None, n_train_itr=None, l2_reg=0.0, vi_target=None):
        """reference: sqair/model.py:150-168.  Returns (target, grads_and_vars): the VIMCO target (already divided
        by T, plus the l2 term) from the fused ELBO kernel and, when an optimiser is given, the gradients of every
        trainable variable from the HIP backward pass as a list of (gradient tensor, variable name) — the
        reference's ``opt.compute_gradients(target)``; the tensors are views of ``core.flat_grad``.
        ``opt.apply_gradients(gvs)`` (sqair_amd.train.Optimizer; learning rate from the flags' schedule unless given)
        performs the update."""
        core = self.core
        if vi_target is not None:
            if vi_target not in ("vimco", "iwae", "reinforce"):
                raise ValueError("vi_target is 'vimco' (alias 'iwae', the reference's Model.VI_TARGETS name for it) or 'reinforce'")
            vi_target = "vimco" if vi_target == "iwae" else vi_target   # canonical name first: the checks below compare with it
        if self.k_particles == 1 and (vi_target or getattr(core, "vi_target", "vimco")) == "vimco":
            # targets.py:55 divides by k_particles - 1: the reference's target (and this library's) is NaN with one particle
            raise ValueError("the VIMCO control variate needs k_particles >= 2 (sqair/targets.py:55 divides by k_particles - 1); "
                             "with one particle use make_target(..., vi_target='reinforce')")
        if vi_target is not None and vi_target != getattr(core, "vi_target", "vimco"):
            core.set_vi_target(vi_target)   # (`reinforce`: targets.py:78-89; the reference's own make_target always takes vimco)
            self._ran = False
        if opt is None:
            if not self._ran:
                self.run()
            target = self.vimco_target
            if l2_reg != 0.0:
                target = target + l2_reg * 0.5 * (core.flat ** 2).sum()
            return target, None
        with core.on_stream():
            if not self._ran:
                core.draw_noise()
            core.grad_step(use_graph=self._use_graph)
            if self.debug:
                self._debug_checks(train=True)
                core.check_finite(core.flat_grad, "flat gradient of the VIMCO target")
            if l2_reg != 0.0:
                core.check(core.lib.sqair_add_l2_grad(
                    core.handle, core.flat.data_ptr(), core.flat_grad.data_ptr(), core.n_params, float(l2_reg),
                    core._stream()), "sqair_add_l2_grad")
            self._collect()
            target = self.vimco_target
            if l2_reg != 0.0:
                target = target + l2_reg * 0.5 * (core.flat ** 2).sum()
        core.stream.synchronize()
        self._ran = True
        gvs = [(g, name) for name, g in core.grads_by_name().items()]
        assert len(gvs) == len(core.spec)
        return target, gvs

    def img_summaries(self):
        """reference: sqair/model.py:207-214 -> uint8 reconstructions / inputs of the first frame."""
        recs = (self.resampled_canvas.clamp(0.0, 1.0) * 255.0).round().to(torch.uint8)
        return dict(reconstructions=recs[0], inputs=(self.obs[0] * 255.0).round().to(torch.uint8))


def load(img, coords=None, num=None, mean_img=None, debug=False, F=None, params=None, seed=0, device="cuda:0",
         outputs="all"):
    """Model factory with the reference's signature (sqair/configs/mlp_mnist_model.py:74):
    ``img`` float32 [T,B,H,W] in [0,1]; ``coords`` [T,B,n_obj,4] (carried, unused by the maths);
    ``num`` [T,B,n_max+1] prefix-ones presence for the step accuracy; ``mean_img`` [H,W].
    Hyper-parameters come from the flags (``F`` defaults to the global ``sqair_amd.flags.FLAGS``)."""
    F = F if F is not None else FLAGS
    img = torch.as_tensor(img, dtype=torch.float32)
    if img.dim() == 5:
        img = img[..., 0]
    hw = (int(img.shape[2]), int(img.shape[3]))
    core = SqairCore(F, hw, device=device)
    if params is None:
        params = init_params(F, hw, seed=seed, mean_img=mean_img)
    core.set_params(params)
    return Model(img, coords, core, int(F.k_particles), presence=num, debug=debug, outputs=outputs)
