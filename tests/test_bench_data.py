"""bench.py's synthetic data (SURVEY 8d generator): determinism, value domain, and that the extra query batches of the
`fresh_queries` measurement come from the distribution the timed batch comes from (same cluster means)."""
import numpy as np

from rii_amd import bench_data as bd


def test_sift_like_is_deterministic_and_sift_valued():
    a = bd.sift_like(n_base=300, n_train=200, n_query=50, D=32, seed=7, n_clusters=16)
    b = bd.sift_like(n_base=300, n_train=200, n_query=50, D=32, seed=7, n_clusters=16)
    for x, y in zip(a, b):
        assert x.dtype == np.float32 and np.array_equal(x, y)
        assert x.min() >= 0 and x.max() <= 255 and np.array_equal(x, np.rint(x))      # non-negative integers, like SIFT
    assert a[0].shape == (300, 32) and a[1].shape == (200, 32) and a[2].shape == (50, 32)


def test_more_queries_shares_the_cluster_means():
    D, K = 16, 8
    base, _, _ = bd.sift_like(n_base=4000, n_train=1, n_query=1, D=D, seed=11, n_clusters=K)
    more = bd.more_queries(2000, D=D, seed=11, n_clusters=K)
    assert more.shape == (2000, D) and np.array_equal(more, bd.more_queries(2000, D=D, seed=11, n_clusters=K))
    assert not np.array_equal(more, bd.more_queries(2000, D=D, seed=11, n_clusters=K, stream=2))
    means = np.random.default_rng(11).random((K, D), dtype=np.float32) * 128.0
    # every vector of either set lies near one of the SAME K means (noise sigma 24 in 16 dimensions: distance ~ 96 +- 17)
    for x in (base, more):
        d = np.sqrt(((x[:, None, :] - means[None]) ** 2).sum(-1)).min(1)
        assert 60 < d.mean() < 130 and d.max() < 250


def test_recall_at_r():
    ids = np.array([[3, 4], [9, 1], [5, 5]])
    gt = np.array([3, 1, 7])
    assert bd.recall_at_r(ids, gt, 1) == 1.0 / 3.0
    assert bd.recall_at_r(ids, gt, 2) == 2.0 / 3.0


def test_sift1m_dir_files_are_read_in_the_reference_harness_format(tmp_path, monkeypatch):
    """`$SIFT1M_DIR` support was untested.  The real set cannot be downloaded here, so the files are written in the
    format the reference's harness reads (examples/benchmark/util.py:5-32: every row = an int32 dimension header + d values;
    .fvecs float32, .ivecs int32) and must come back as the (base, train, query) bench.py measures on -- sliced to the requested
    sizes, float32, headers gone."""
    rng = np.random.default_rng(3)
    D = 128

    def write_fvecs(name, x):
        rows = np.empty((x.shape[0], D + 1), np.int32)
        rows[:, 0] = D
        rows[:, 1:] = x.astype(np.float32).view(np.int32)
        rows.tofile(str(tmp_path / name))

    base = np.rint(rng.random((50, D)) * 255).astype(np.float32)
    learn = np.rint(rng.random((30, D)) * 255).astype(np.float32)
    query = np.rint(rng.random((20, D)) * 255).astype(np.float32)
    write_fvecs("sift_base.fvecs", base)
    write_fvecs("sift_learn.fvecs", learn)
    write_fvecs("sift_query.fvecs", query)
    gt = rng.integers(0, 50, size=(20, 100)).astype(np.int32)
    g = np.empty((20, 101), np.int32)
    g[:, 0] = 100
    g[:, 1:] = gt
    g.tofile(str(tmp_path / "sift_groundtruth.ivecs"))
    monkeypatch.setenv("SIFT1M_DIR", str(tmp_path))
    b, t, q = bd.sift_like(n_base=40, n_train=25, n_query=10, D=D)
    assert b.dtype == np.float32 and np.array_equal(b, base[:40]) and np.array_equal(t, learn[:25]) and np.array_equal(q, query[:10])
    assert np.array_equal(bd.read_ivecs(str(tmp_path / "sift_groundtruth.ivecs")), gt)
    assert np.array_equal(bd.read_fvecs(str(tmp_path / "sift_base.fvecs"), count=7), base[:7])
    # another dimension falls back to the generator (the real files are 128-dimensional)
    b2, _, _ = bd.sift_like(n_base=5, n_train=1, n_query=1, D=32, n_clusters=4)
    assert b2.shape == (5, 32)
