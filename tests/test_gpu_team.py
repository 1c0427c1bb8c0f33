"""The walk on a team of wavefronts per query (small launches, long rows) against the one-wavefront register beam: at
equal total width per step it is the same walk (hnsw.rs:333-368, 420-511) — same ids, same keys bit for bit, same
expansion counts — for every list size, row shape and metric; at its own default width (four nodes per wavefront and
step) recall stays within 0.5 pt of the oracle's at equal ef on the same graph; tombstones, high-degree nodes and a
visited table far too small change nothing."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def data(n, d, seed, kind="clustered"):
    rng = np.random.default_rng(seed)
    if kind == "uniform":
        return rng.uniform(-1, 1, size=(n, d)).astype(np.float32)
    centres = rng.uniform(0, 128, size=(max(8, n // 200), d))
    x = centres[rng.integers(0, len(centres), size=n)] + rng.normal(0, 12, size=(n, d))
    return np.clip(x, 0, None).astype(np.float32)


def search(h, q, k, ef, monkeypatch, **env):
    for key, v in env.items(): monkeypatch.setenv(key, str(v))
    out = h.search_batch(q, k, ef_search=ef, with_stats=True)
    for key in env: monkeypatch.delenv(key)
    return out


def recall(a, an, e, en, k):
    return sum(len(set(x["id"][:n].tolist()) & set(y["id"][:m].tolist())) for x, n, y, m in zip(a, an, e, en)) / (len(a) * k)


@pytest.mark.parametrize("metric,dim,n,ef", [(2, 128, 30000, 144), (0, 960, 12000, 384), (1, 96, 20000, 64), (2, 1536, 6000, 100),
                                             (0, 20, 20000, 300), (1, 260, 8000, 500), (2, 64, 5000, 17)])
def test_a_team_at_equal_width_is_the_one_wavefront_walk(vdb, monkeypatch, metric, dim, n, ef):
    nq, k = 250, 10
    x = data(n, dim, 7 + dim); q = data(nq, dim, 8 + dim)
    h = vdb.HNSWIndex(metric, 16, 64, 16, dim, seed=3)
    h.insert_batch(np.arange(n, dtype=np.uint64) * 3 + 1, x)
    for gone in range(0, n, 41): h.mark_deleted(gone * 3 + 1)                 # tombstones inside the steps (hnsw.rs:475)
    for width in (4, 1):
        ref, refn, rst = search(h, q, k, ef, monkeypatch, VDB_TEAM=0, VDB_BEAM_WIDTH=width)
        for team in (2, 4):
            got, gotn, st = search(h, q, k, ef, monkeypatch, VDB_TEAM=team, VDB_BEAM_WIDTH=width)
            assert np.array_equal(refn, gotn)
            assert np.array_equal(ref["sort_key"].view(np.uint32), got["sort_key"].view(np.uint32)), (team, width)
            assert np.array_equal(ref["raw_score"].view(np.uint32), got["raw_score"].view(np.uint32))
            assert np.array_equal(ref["id"], got["id"]), (team, width, np.argwhere(ref["id"] != got["id"])[:4])
            assert np.array_equal(rst["expanded"], st["expanded"]), (team, width)   # the same walk, node for node
            assert not st["flags"].any()
    # a visited table of 128 tags forgets nearly every id: more rows are scored, the results stay
    ref, refn, rst = search(h, q, k, ef, monkeypatch, VDB_TEAM=0, VDB_BEAM_WIDTH=4)
    tiny, tinyn, tst = search(h, q, k, ef, monkeypatch, VDB_TEAM=4, VDB_BEAM_WIDTH=4, VDB_TAG_LOG2=7)
    assert np.array_equal(tinyn, refn) and np.array_equal(tiny["id"], ref["id"]) and np.array_equal(tst["expanded"], rst["expanded"])
    assert np.array_equal(tiny["sort_key"].view(np.uint32), ref["sort_key"].view(np.uint32))
    assert tst["rows_scored"].mean() > rst["rows_scored"].mean()


@pytest.mark.parametrize("metric", [0, 1, 2])
def test_a_team_at_its_own_width_keeps_the_oracles_recall(ora, vdb, monkeypatch, metric):
    n, d, nq, k = 6000, 64, 300, 10
    x = data(n, d, 31 + metric); q = data(nq, d, 77)
    ho = ora.HNSWIndex(metric, 16, 64, 16, d, seed=31)
    for i in range(n): ho.insert(i * 3 + 1, x[i])
    exact, exact_n = ho.exact_knn_batch(q, k)
    h = vdb.HNSWIndex(metric, 16, 64, 16, d, seed=1); h.import_graph(ho.export_graph())
    for ef in (32, 64, 200, 320, 512):
        ref, refn = ho.search_batch(q, k, ef_search=ef)
        for team in (2, 4):
            got, gotn, st = search(h, q, k, ef, monkeypatch, VDB_TEAM=team)       # 4 nodes per wavefront and step
            assert np.array_equal(refn, gotn)
            assert recall(got, gotn, exact, exact_n, k) >= recall(ref, refn, exact, exact_n, k) - 0.005, (ef, team)
            assert all(np.all(np.diff(got["sort_key"][i][: gotn[i]]) >= 0) for i in range(nq))
            assert all(len(set(got["id"][i][: gotn[i]].tolist())) == gotn[i] for i in range(nq))
            for qi in range(0, nq, 11):                                           # every distance within 1e-5 of the strict score
                ids = got["id"][qi][: gotn[qi]]
                rows = x[(ids.astype(np.int64) - 1) // 3]
                strict = np.array([ora.score(ho.maybe_normalize(q[qi]), ho.maybe_normalize(r), metric) for r in rows], np.float32)
                scale = {0: np.ones(len(ids)), 1: np.linalg.norm(q[qi]) * np.linalg.norm(rows, axis=1), 2: np.abs(strict)}[metric]
                assert np.all(np.abs(got["raw_score"][qi][: gotn[qi]] - strict) <= 1e-5 * np.maximum(scale, 1e-6))


def test_a_team_walks_high_degree_nodes_and_answers_one_query(ora, vdb, monkeypatch):
    """Nodes with more than 62 neighbours (overflow chunks of the adjacency block, visited in further rounds of a step),
    strictly best-first: the oracle's ids; and the single-query call (the reference's own call shape) takes the team."""
    metric, n, d, nq, k = 2, 5000, 16, 200, 10
    rng = np.random.default_rng(3)
    x = rng.normal(0, 1, size=(n, d)).astype(np.float32); x[:40] *= 0.05
    ho = ora.HNSWIndex(metric, 16, 64, 16, d, seed=9)
    for i in range(n): ho.insert(i * 3 + 1, x[i])
    g = ho.export_graph()
    assert (np.diff(g["layers"][0][1].astype(np.int64)) > 62).sum() >= 5
    q = rng.normal(0, 1, size=(nq, d)).astype(np.float32)
    h = vdb.HNSWIndex(metric, 16, 64, 16, d, beam_width=1); h.import_graph(g)
    for ef in (64, 200, 400):
        ref, refn = ho.search_batch(q, k, ef_search=ef)
        for team in (2, 4):
            got, gotn, st = search(h, q, k, ef, monkeypatch, VDB_TEAM=team)
            assert np.array_equal(refn, gotn)
            assert np.mean(ref["id"] == got["id"]) >= 0.995
            assert np.allclose(ref["sort_key"], got["sort_key"], rtol=1e-5)
    one = h.search(q[3], k)
    ref1 = ho.search(q[3], k)
    assert [r.id for r in one] == [r.id for r in ref1]
