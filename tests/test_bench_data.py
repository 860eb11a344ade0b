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
