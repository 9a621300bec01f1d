"""Host-side logic that needs no GPU: index streams, row sharding, operator recognition."""
import numpy as np
import pytest


def test_index_stream_is_chunking_independent(ciao):
    a = ciao.IndexStream(7)
    b = ciao.IndexStream(7)
    whole = a.rand_indices(1000, 513)
    parts = np.concatenate([b.rand_indices(1000, k) for k in (1, 2, 10, 500)])
    assert np.array_equal(whole, parts)
    assert whole.min() >= 0 and whole.max() < 1000 and whole.dtype == np.int64
    assert not np.array_equal(ciao.IndexStream(8).rand_indices(1000, 513), whole)


def test_index_stream_uniformity_and_range(ciao):
    s = ciao.IndexStream(0)
    N = 10
    x = s.rand_indices(N, 200_000)
    counts = np.bincount(x, minlength=N)
    assert counts.min() > 19_000 and counts.max() < 21_000
    big = s.rand_indices(80_000_000, 1000)
    assert big.min() >= 0 and big.max() < 80_000_000


def test_randperm_and_sampling_without_replacement(ciao):
    s = ciao.IndexStream(3)
    for n in (1, 2, 17, 1000):
        p = s.randperm(n)
        assert sorted(p.tolist()) == list(range(n))
    for N, r in ((8, 1), (8, 3), (8, 8), (1000, 10), (10**7, 4096)):
        x = s.sample_without_replacement(N, r)
        assert x.shape == (r,) and len(set(x.tolist())) == r and x.min() >= 0 and x.max() < N


def test_fixed_stream_replays(ciao):
    f = ciao.FixedStream(indices=[3, 1, 2, 0], perms=[[1, 0]], samples=[[2, 3]])
    assert f.rand_indices(4, 3).tolist() == [3, 1, 2]
    assert f.rand_indices(4, 1).tolist() == [0]
    with pytest.raises(IndexError):
        f.rand_indices(4, 1)
    assert f.randperm(2).tolist() == [1, 0] and f.sample_without_replacement(4, 2).tolist() == [2, 3]


def test_shard_rows_partitions_exactly():
    import ciao_loader
    ciao_loader.load()
    from ciaoalgorithms_jl_amd.parallel import shard_rows
    for N in (0, 1, 7, 8, 1000, 80_000_000):
        for world in (1, 2, 3, 8):
            spans = [shard_rows(N, r, world) for r in range(world)]
            assert spans[0][0] == 0 and sum(n for _, n in spans) == N
            for (r0, n0), (r1, _) in zip(spans, spans[1:]):
                assert r0 + n0 == r1
            assert max(n for _, n in spans) - min(n for _, n in spans) <= 1
    assert shard_rows(80_000_000, 3, 8) == (30_000_000, 10_000_000)


def test_cyclic_ownership_covers_every_row_once():
    import ciao_loader
    ciao_loader.load()
    from ciaoalgorithms_jl_amd.parallel import shard_rows_cyclic
    for N in (0, 1, 7, 64, 1001):
        for world in (1, 2, 3, 8):
            counts = [shard_rows_cyclic(N, r, world) for r in range(world)]
            assert sum(counts) == N and counts == [len(range(r, N, world)) for r in range(world)]


def test_static_batches_match_the_reference_layout():
    import ciao_loader
    ciao_loader.load()
    from ciaoalgorithms_jl_amd.solvers import _static_batch
    N, r = 10, 4   # Finito_basic.jl:52-58: [0..3], [4..7], remainder [8, 9]
    assert [_static_batch(N, r, j).tolist() for j in range(3)] == [[0, 1, 2, 3], [4, 5, 6, 7], [8, 9]]


def test_solver_constructors_validate_like_the_reference():
    import ciao_loader
    ciao_loader.load()
    from ciaoalgorithms_jl_amd import solvers as S
    s = S.SVRG()                                    # SVRG(; kw...) = SVRG(Float64; kw...)  (SVRG.jl:113)
    assert s.maxit == 10000 and s.freq == 1000 and s.m is None and s.plus is False and s.γ is None
    s = S.SAGA(np.float32, gamma=0.5)
    assert s.γ == 0.5 and s.SAG_flag is False and S.SAG(np.float32).SAG_flag is True
    f = S.Finito()
    assert (f.sweeping, f.LFinito, f.adaptive, f.minibatch, f.maxit, f.freq, f.α) == (1, False, False, (False, 1), 10000, 10000, 0.999)
    for bad in (dict(γ=0.0), dict(γ=-1.0), dict(maxit=0), dict(freq=0)):
        with pytest.raises(AssertionError):
            S.SVRG(**bad)
    with pytest.raises(AssertionError):
        S.Finito(tol=0.0)
    with pytest.raises(TypeError):
        S.SVRG(γ=0.1, gamma=0.1)


def test_operator_descriptions_validate():
    import ciao_loader
    ciao_loader.load()
    from ciaoalgorithms_jl_amd import operators as ops
    with pytest.raises(ValueError):
        ops.NormL1(-1.0)
    with pytest.raises(ValueError):
        ops.IndBox(1.0, 0.0)
    with pytest.raises(ValueError):
        ops.LogisticLoss([0.5])
    with pytest.raises(ValueError):
        ops.LeastSquares(np.zeros((2, 3)), np.zeros(3))
    f = ops.LeastSquares(np.ones((1, 3)), [2.0], 6.0)
    assert f.A.shape == (1, 3) and f.b.shape == (1,) and f.lam == 6.0
