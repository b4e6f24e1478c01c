"""Host-side checks that need no GPU: the C-ABI library loads, exports every symbol include/sqair_hip.h
declares, and its parameter table agrees with the Python inventory."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from sqair_amd import _capi
from sqair_amd.flags import make_flags
from sqair_amd.model import make_config
from sqair_amd.params import param_offsets, param_spec


def _header_functions(repo_root):
    txt = open(os.path.join(repo_root, "include", "sqair_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(sqair_[a-z_0-9]+)\s*\(", txt)))


def test_library_loads_and_exports_header_symbols(repo_root):
    lib = _capi.lib()
    names = _header_functions(repo_root)
    assert len(names) >= 18
    for n in names:
        assert hasattr(lib, n), "missing export " + n
    assert set(names) == set(_capi.EXPORTED_SYMBOLS), "ctypes prototypes and header disagree"
    assert lib.sqair_abi_version() == _capi.ABI_VERSION == 2
    assert lib.sqair_build_flags() == b"product"


def test_build_id_is_compiled_in_and_a_stale_binary_is_refused(tmp_path):
    """The id a library reports is the hash of the sources IT was compiled from (csrc/build.py passes it to hipcc), not a hash
    taken from disk at run time: a binary that does not belong to the sources beside it is refused by the binding, and the
    measurement tools (bench.py, tools/timeline.py) get their `build_id` from the loaded library."""
    from sqair_amd.csrc import build as B
    assert _capi.build_id() == _capi.source_id() == B.binary_id(_capi.LIB_PATH)
    assert re.fullmatch(r"[0-9a-f]{16}", _capi.build_id())
    # a deliberately stale copy: same code, another id compiled in (both occurrences: the marker and the exported string)
    blob = open(_capi.LIB_PATH, "rb").read()
    good = _capi.build_id().encode()
    assert blob.count(good) >= 2
    stale = tmp_path / "libsqair_hip.so"
    stale.write_bytes(blob.replace(good, b"0123456789abcdef"))
    assert B.binary_id(str(stale)) == "0123456789abcdef"
    with pytest.raises(_capi.StaleLibraryError, match="compiled from sources 0123456789abcdef but the sources on disk are " + good.decode()):
        _capi.lib(str(stale))
    assert _capi.lib(str(stale), allow_stale=True).sqair_build_id() == b"0123456789abcdef"
    assert _capi.build_id(str(stale)) == "0123456789abcdef"   # the LIBRARY's value, whatever the disk says
    # csrc/build.py rebuilds on an id mismatch, not on modification times
    assert B.binary_id(str(tmp_path / "absent.so")) is None


@pytest.mark.parametrize("N,hw,cell", [(3, (50, 50), "GRU"), (4, (50, 50), "GRU"), (6, (50, 50), "GRU"), (4, (128, 128), "GRU"),
                                       (3, (50, 50), "LSTM"), (3, (50, 50), "GRU/LSTM"), (4, (50, 50), "LSTM/LSTM"),
                                       (3, (50, 50), "GRU/GRU/LSTM"), (4, (50, 50), "LSTM/LSTM/LSTM"),
                                       (3, (50, 50), "GRU/GRU/GRU"), (3, (50, 50), "VanillaRNN/VanillaRNN/LSTM"),
                                       (4, (50, 50), "VanillaRNN/LSTM/VanillaRNN")])
def test_param_table_matches_python_spec(N, hw, cell):
    lib = _capi.lib()
    cs = cell.split("/") + ["GRU", "VanillaRNN"][len(cell.split("/")) - 1:]
    F = make_flags(n_steps_per_image=N, time_transition=cs[0], prior_transition=cs[1], transition=cs[2])
    cfg = make_config(F, hw)
    h = C.c_void_p()
    assert lib.sqair_create(C.byref(cfg), C.byref(h)) == 0
    try:
        spec = param_spec(F, hw)
        off, total = param_offsets(spec)
        assert lib.sqair_param_count(h) == total
        assert lib.sqair_param_entries(h) == len(spec)
        for i, (name, shape, _, _) in enumerate(spec):
            cname, coff, cnum = C.c_char_p(), C.c_int64(), C.c_int64()
            assert lib.sqair_param_entry(h, i, C.byref(cname), C.byref(coff), C.byref(cnum)) == 0
            assert cname.value.decode() == name
            assert coff.value == off[name][0]
            assert cnum.value == (int(np.prod(shape)) if len(shape) else 1)
        if N == 3 and hw == (50, 50) and cell == "GRU":
            assert total == 2951522  # reference notebooks/play.ipynb:362
        assert lib.sqair_packed_bytes(h) > total * 4
        assert lib.sqair_workspace_bytes(h, 10, 32) > 0
        assert lib.sqair_noise_width(h) == 55
    finally:
        lib.sqair_destroy(h)


def test_create_rejects_bad_config():
    lib = _capi.lib()
    F = make_flags()
    cfg = make_config(F, (50, 50))
    cfg.n_steps_per_image = 0
    h = C.c_void_p()
    assert lib.sqair_create(C.byref(cfg), C.byref(h)) != 0
    cfg = make_config(F, (50, 50))
    cfg.n_hidden = 72          # not 32 * n_units: n_hidden // 2 must be a whole number of 16-column tiles' worth of units
    assert lib.sqair_create(C.byref(cfg), C.byref(h)) != 0
    with pytest.raises(ValueError):
        make_config(make_flags(n_units=0), (50, 50))
    cfg = make_config(F, (37, 41))   # any frame size (H * W not a multiple of 4 goes through a padded copy)
    assert lib.sqair_create(C.byref(cfg), C.byref(h)) == 0
    lib.sqair_destroy(h)


def test_flag_errors_mirror_reference():
    with pytest.raises(ValueError):
        make_config(make_flags(prop_prior_type="bogus"), (50, 50))   # propagate.py:42-43
    with pytest.raises(ValueError):
        make_config(make_flags(disc_prior_type="bogus"), (50, 50))   # sqair_modules.py:224
    with pytest.raises(ValueError):
        make_flags(not_a_flag=1)


@pytest.mark.parametrize("flags", [dict(n_what=100), dict(n_what=128, n_units=16), dict(n_steps_per_image=12), dict(n_units=10),
                                   dict(n_units=13, time_transition="LSTM", prior_transition="LSTM", transition="LSTM"),
                                   dict(n_steps_per_image=14, n_what=128)])
def test_wide_library_accepts_the_rest_of_the_flag_range(flags):
    """Beyond what the product library is laid out for (n_what <= 50, 8 slots, n_units <= 8) the SAME sources compiled with
    -DSQAIR_WIDE serve the configuration through the same C-ABI (reference flags take any value:
    sqair/common_model_flags.py:32-56, configs/mlp_mnist_model.py:42-52); the product library says no instead of misbehaving."""
    F = make_flags(**flags)
    cfg = make_config(F, (50, 50))
    path = _capi.lib_path_for(cfg.n_what, cfg.n_steps_per_image, cfg.n_hidden)
    assert path == _capi.WIDE_LIB_PATH
    h = C.c_void_p()
    assert _capi.lib().sqair_create(C.byref(cfg), C.byref(h)) != 0
    lib = _capi.lib(path)
    assert lib.sqair_build_flags() == b"wide" and lib.sqair_abi_version() == _capi.ABI_VERSION
    assert lib.sqair_create(C.byref(cfg), C.byref(h)) == 0
    try:
        spec = param_spec(F, (50, 50))
        off, total = param_offsets(spec)
        assert lib.sqair_param_count(h) == total and lib.sqair_param_entries(h) == len(spec)
        for i, (name, shape, _, _) in enumerate(spec):
            cname, coff, cnum = C.c_char_p(), C.c_int64(), C.c_int64()
            assert lib.sqair_param_entry(h, i, C.byref(cname), C.byref(coff), C.byref(cnum)) == 0
            assert (cname.value.decode(), coff.value, cnum.value) == (name, off[name][0], int(np.prod(shape)) if len(shape) else 1)
        assert lib.sqair_noise_width(h) == 4 + int(F.n_what) + 1
        assert lib.sqair_workspace_bytes(h, 3, 2) > 0 and lib.sqair_backward_bytes(h, 3, 2) > 0
    finally:
        lib.sqair_destroy(h)


def test_limits_of_both_builds():
    lib, wide = _capi.lib(), _capi.lib(_capi.WIDE_LIB_PATH)
    h = C.c_void_p()
    for flags, ok_product, ok_wide in [(dict(n_what=50, n_steps_per_image=8, n_units=8), True, True),
                                       (dict(n_what=51), False, True), (dict(n_steps_per_image=9), False, True),
                                       (dict(n_units=9), False, True), (dict(n_what=129), False, False),
                                       (dict(n_steps_per_image=17), False, False), (dict(n_units=17), False, False),
                                       (dict(k_particles=256), True, True), (dict(k_particles=257), False, False),
                                       (dict(n_steps_per_image=16, n_what=128), False, False)]:   # (the adjoint's LDS staging)
        cfg = make_config(make_flags(**flags), (50, 50))
        for l, ok in ((lib, ok_product), (wide, ok_wide)):
            rc = l.sqair_create(C.byref(cfg), C.byref(h))
            assert (rc == 0) == ok, (flags, l.sqair_build_flags(), rc)
            if rc == 0:
                l.sqair_destroy(h)


def test_run_time_options_host_side():
    """sqair_set_option without a GPU: the in-launch slot chain lays the inference workspace out like the training tape for the
    configurations it serves (shipped cells, n_hidden 256, up to 320 particle rows) and leaves the others alone; the wide build
    refuses it; vi_target takes 0 / 1."""
    lib = _capi.lib()

    def handle(library, **flags):
        F = make_flags(k_particles=5, n_steps_per_image=4, **flags)
        cfg = make_config(F, (50, 50))
        h = C.c_void_p()
        assert library.sqair_create(C.byref(cfg), C.byref(h)) == 0
        return h
    h = handle(lib)
    base = lib.sqair_workspace_bytes(h, 10, 32)
    train = lib.sqair_train_workspace_bytes(h, 10, 32)
    assert lib.sqair_set_option(h, b"slot_chain", 1) == 0
    on = lib.sqair_workspace_bytes(h, 10, 32)
    assert on > 5 * base and on >= train          # per-slot buffers kept apart + the launches' control blocks
    assert lib.sqair_workspace_bytes(h, 10, 256) == _fresh_bytes(lib, 10, 256)   # 1280 rows: the chain keeps out
    assert lib.sqair_set_option(h, b"slot_chain", 0) == 0
    assert lib.sqair_workspace_bytes(h, 10, 32) == base
    assert lib.sqair_set_option(h, b"vi_target", 1) == 0 and lib.sqair_set_option(h, b"vi_target", 0) == 0
    assert lib.sqair_set_option(h, b"vi_target", 7) == -2 and b"vi_target" in lib.sqair_last_error(h)
    lib.sqair_destroy(h)
    h = handle(lib, time_transition="LSTM")
    base = lib.sqair_workspace_bytes(h, 10, 32)
    assert lib.sqair_set_option(h, b"slot_chain", 1) == 0
    assert lib.sqair_workspace_bytes(h, 10, 32) == base      # LSTM temporal cell: one launch per op
    lib.sqair_destroy(h)
    wide = _capi.lib(_capi.WIDE_LIB_PATH)
    h = handle(wide)
    assert wide.sqair_set_option(h, b"slot_chain", 1) == -2 and b"wide build" in wide.sqair_last_error(h)
    wide.sqair_destroy(h)


def _fresh_bytes(lib, T, B):
    F = make_flags(k_particles=5, n_steps_per_image=4)
    cfg = make_config(F, (50, 50))
    h = C.c_void_p()
    assert lib.sqair_create(C.byref(cfg), C.byref(h)) == 0
    n = lib.sqair_workspace_bytes(h, T, B)
    lib.sqair_destroy(h)
    return n
