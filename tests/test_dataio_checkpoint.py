"""Host-side rows of SURVEY.md 8(f): dataset pickle layout / minibatch feed (rank 2) and checkpoint interchange by TF
variable name (rank 3).  No GPU."""
import numpy as np
import pytest

from sqair_amd import checkpoint as ck
from sqair_amd import dataio
from sqair_amd.data import make_sequences
from sqair_amd.flags import make_flags
from sqair_amd.params import flatten_params, init_params, param_spec


def _dataset(n=12, T=5):
    d = make_sequences(n, T=T, canvas=(50, 50), n_objects=(0, 2), seed=9)
    # the reference stores nums once per sequence ([1, N, n_max+1]) and coords without the padding object
    return dict(imgs=d["imgs"], nums=d["nums"][:1].astype(np.uint8), coords=d["coords"][..., :2, :].astype(np.float32),
                labels=np.zeros((n, 2), dtype=np.int64))


def test_dataset_pickle_round_trip_and_processing(tmp_path):
    data = _dataset()
    path = str(tmp_path / "seq_mnist_validation.pickle")
    dataio.save_dataset(path, data)
    got = dataio.load_dataset(path)
    assert got["imgs"].dtype == np.float32 and got["imgs"].max() <= 1.0 and got["imgs"].shape == (5, 12, 50, 50)
    assert np.array_equal(np.round(got["imgs"] * 255).astype(np.uint8), data["imgs"])
    assert got["nums"].dtype == np.float32
    dataio.process_data(got, n_timesteps=4)
    assert got["imgs"].shape[0] == 4 and got["coords"].shape[:2] == (4, 12)
    assert got["coords"].shape[-2] == got["nums"].shape[-1]          # padded with zero boxes to n_max + 1 objects
    assert np.all(got["coords"][..., -1, :] == 0)


def test_minibatch_feed_matches_reference_semantics():
    data = dataio.process_data(_dataset(n=10, T=5))
    data["imgs"] = data["imgs"].astype(np.float32) / 255.0
    feed = dataio.MinibatchFeed(data, batch_size=4, shuffle=False)
    starts = [int(feed.indices()[0]) for _ in range(5)]
    assert starts == [0, 4, 0, 4, 0]                                 # itertools.cycle(range(0, n - b + 1, b)): tail never visited
    b = feed.next()
    assert b["imgs"].shape == (5, 4, 50, 50) and b["nums"].shape == (5, 4, 3) and b["coords"].shape == (5, 4, 3, 4)
    assert b["labels"].shape == (4, 2)
    assert np.array_equal(b["nums"][0], b["nums"][4])                # nums stored once, tiled over time
    sh = dataio.MinibatchFeed(data, batch_size=6, shuffle=True, seed=0)
    idx = sh.indices()
    assert idx.shape == (6,) and idx.min() >= 0 and idx.max() < 10   # with replacement, like np.random.choice
    cur = dataio.MinibatchFeed(data, batch_size=4, shuffle=False, seq_len=2, stage_itr=100)
    assert cur.next(step=0)["imgs"].shape[0] == 2
    assert cur.next(step=250)["imgs"].shape[0] == 4
    assert cur.next(step=10 ** 6)["imgs"].shape[0] == 5              # capped at the data length
    with pytest.raises(ValueError):
        dataio.MinibatchFeed(data, batch_size=11, shuffle=False)


def test_checkpoint_is_keyed_by_tf_variable_names(tmp_path):
    F = make_flags()
    hw = (50, 50)
    spec = param_spec(F, hw)
    P = init_params(F, hw, seed=3, jitter=0.1)
    tfd = ck.to_tf_dict(P, F, hw)
    assert len(tfd) == len(spec) == 105
    assert sum(int(v.size) for v in tfd.values()) == 2951522         # notebooks/play.ipynb:239-362
    for expected in ("decoder/air_decoder/Variable", "decoder/air_decoder/decoder/mlp/linear_2/w"):
        assert expected in tfd
    back = ck.from_tf_dict(tfd, F, hw)
    assert np.array_equal(flatten_params(back, spec), flatten_params(P, spec))
    # a dump with transposed storage but the right element count is accepted only if it has the variable's size
    bad = dict(tfd)
    bad["decoder/air_decoder/Variable"] = np.zeros(7, dtype=np.float32)
    with pytest.raises(ValueError):
        ck.from_tf_dict(bad, F, hw)
    del bad["decoder/air_decoder/Variable"]
    with pytest.raises(KeyError):
        ck.from_tf_dict(bad, F, hw)
    assert "dec.mean_img" not in ck.from_tf_dict(bad, F, hw, strict=False)
