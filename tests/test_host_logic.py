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


# ---- batch selection for n iterations at a time (solvers._next_batches_packed, sampling.sample_batches) -----------------
@pytest.mark.parametrize("N,r,n", [(50, 20, 40), (1000, 500, 6), (1000, 37, 200), (200_000, 256, 50), (10, 1, 30), (7, 5, 10),
                                    (100, 50, 20), (4, 2, 50), (3, 3, 4)])
def test_sample_batches_equals_separate_draws(ciao, N, r, n):
    """n batches in one call (the library's host helper) == n calls of the interpreted rule: same indices, same stream
    position afterwards -- with heavy collisions (2r = N), r = 1, and the shuffle branch (2r > N)."""
    a, b = ciao.IndexStream(5), ciao.IndexStream(5)
    a.rand_indices(N, 3), b.rand_indices(N, 3)
    A = np.array(a.sample_batches(N, r, n))
    B = np.stack([b.sample_without_replacement(N, r) for _ in range(n)])
    assert np.array_equal(A, B) and a.pos == b.pos
    assert all(len(set(row.tolist())) == r for row in A) and A.min() >= 0 and A.max() < N
    assert np.array_equal(a.rand_indices(N, 5), b.rand_indices(N, 5))


class _FakeF:
    N = N_total = 23
    cyclic = None


class _FakeIt:
    def __init__(self, sweeping, stream, N=23, batch=4):
        self.sweeping, self.stream, self.N, self.batch, self.F = sweeping, stream, N, batch, _FakeF()


class _St:
    def __init__(self, d):
        self.d, self.idxr, self.idx, self.inds = d, 1, 0, np.arange(d, dtype=np.int64)   # Finito_basic.jl:37-40


def _one_step_reference(it, st):
    """Finito_basic.jl:95-108, literally, one iteration (1-based idxr as in the reference)."""
    if it.sweeping == 2:
        st.idxr = st.idxr % st.d + 1
    else:
        if st.idx == st.d:
            st.inds = it.stream.randperm(st.d)
            st.idx = 1
        else:
            st.idx += 1
        st.idxr = int(st.inds[st.idx - 1]) + 1
    lo = it.batch * (st.idxr - 1)
    return np.arange(lo, min(lo + it.batch, it.N), dtype=np.int64)


@pytest.mark.parametrize("sweeping", [2, 3])
@pytest.mark.parametrize("chunks", [[1] * 20, [7, 1, 13, 2], [23]])
def test_static_batch_sequences_in_chunks(ciao, sweeping, chunks):
    """Cyclic (first step = batch 2) and shuffled (first pass = identity order, fresh randperm per pass) batch choice:
    n iterations at a time give the batches of n single iterations, across pass boundaries, ragged last batch included."""
    from ciaoalgorithms_jl_amd import solvers as S
    d = -(-23 // 4)
    it_a, st_a = _FakeIt(sweeping, ciao.IndexStream(3)), _St(d)
    it_b, st_b = _FakeIt(sweeping, ciao.IndexStream(3)), _St(d)
    for n in chunks:
        bptr, bidx = S._next_batches_packed(it_a, st_a, n)
        ref = [_one_step_reference(it_b, st_b) for _ in range(n)]
        assert np.array_equal(bptr, np.concatenate([[0], np.cumsum([len(x) for x in ref])]))
        assert np.array_equal(bidx, np.concatenate(ref))
        assert st_a.idxr == st_b.idxr


def test_local_blocks_agree_with_localise():
    """A contiguous block of global rows is a contiguous block of LOCAL rows under both ownership rules (device.PackedF):
    local_blocks (array arithmetic, feeds the index-free *_blocks entry points) against localise (explicit indices)."""
    import ciao_loader
    ciao_loader.load()
    from ciaoalgorithms_jl_amd.device import PackedF

    class Shard:   # the ownership fields of a PackedF, without a device
        localise = PackedF.localise
        local_blocks = PackedF.local_blocks

        def __init__(self, N, N_total, row0=0, cyclic=None):
            self.N, self.N_total, self.row0, self.cyclic = N, N_total, row0, cyclic

    rng = np.random.default_rng(0)
    N_total = 1003
    for world in (1, 2, 3, 8):
        for rank in range(world):
            base, rem = divmod(N_total, world)
            n = base + (1 if rank < rem else 0)
            row0 = rank * base + min(rank, rem)
            shards = [Shard(n, N_total, row0=row0)] if world > 1 else [Shard(N_total, N_total)]
            if world > 1:
                shards.append(Shard((N_total - rank + world - 1) // world, N_total, cyclic=(rank, world)))
            lo = rng.integers(0, N_total, 200)
            hi = np.minimum(lo + rng.integers(0, 300, 200), N_total)
            for sh in shards:
                first, length = sh.local_blocks(lo, hi)
                for a, b, f, l in zip(lo, hi, first, length):
                    want = sh.localise(np.arange(a, b, dtype=np.int64))
                    assert l == want.size and (l == 0 or (f == want[0] and np.array_equal(want, np.arange(f, f + l)))), (world, rank, a, b)
                    assert 0 <= f and f + l <= sh.N
