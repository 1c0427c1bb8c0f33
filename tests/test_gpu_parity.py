"""GPU parity: the HIP path (through the C ABI) against the CPU oracle on the same seeded inputs and
the SAME graph.  Bars: STRICT arithmetic is bit-exact in ids and scores on continuous random data;
FAST arithmetic matches ids outside near-ties (<= 1e-6 relative key gap) and scores to 1e-5 relative."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

REL = 1e-5          # north-star tolerance for distances
TIE = 1e-6          # relative key gap below which an id swap is a tie, not a mismatch


def data(n, d, seed, kind="uniform"):
    rng = np.random.default_rng(seed)
    if kind == "uniform":
        return rng.uniform(-1, 1, size=(n, d)).astype(np.float32)
    centres = rng.uniform(0, 128, size=(32, d))
    x = centres[rng.integers(0, 32, size=n)] + rng.normal(0, 12, size=(n, d))
    return np.clip(x, 0, None).astype(np.float32)


def rel_close(a, b, tol=REL):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    return np.all(np.abs(a - b) <= tol * np.maximum(1e-6, np.maximum(np.abs(a), np.abs(b))))


@pytest.mark.parametrize("dim", [1, 3, 4, 5, 32, 96, 128, 130, 324, 768, 960, 1000, 1920, 2048])
@pytest.mark.parametrize("metric", [0, 1, 2])
def test_score_scan_strict_bitwise_and_fast_close(ora, vdb, dim, metric):
    q = data(9, dim, 1); b = data(257, dim, 2)
    ref = ora.score_scan(q, b, metric)
    got = vdb.score_scan(q, b, metric, arith=vdb.Arith.Strict)
    assert np.array_equal(ref.view(np.uint32), got.view(np.uint32)), f"strict scan differs (max abs {np.abs(ref-got).max()})"
    fast = vdb.score_scan(q, b, metric, arith=vdb.Arith.Fast)
    # relative to the magnitude the sum is built from: 1 for a cosine distance, |q||b| for a dot
    # product (which cancels towards 0), the distance itself for L2
    if metric == 0: scale = np.ones_like(ref)
    elif metric == 1: scale = np.linalg.norm(q, axis=1)[:, None] * np.linalg.norm(b, axis=1)[None, :]
    else: scale = np.abs(ref)
    assert np.all(np.abs(fast - ref) <= REL * np.maximum(scale, 1e-6)), (np.abs(fast - ref) / np.maximum(scale, 1e-6)).max()


def test_score_single_pair_matches_oracle_bitwise(ora, vdb):
    rng = np.random.default_rng(5)
    for metric in (0, 1, 2):
        for d in (2, 3, 7, 64):
            a = rng.normal(size=d).astype(np.float32); b = rng.normal(size=d).astype(np.float32)
            assert np.float32(ora.score(a, b, metric)).view(np.uint32) == np.float32(vdb.score(a, b, metric)).view(np.uint32)


def build_oracle(ora, metric, n, d, seed, m=16, ef=64, kind="uniform"):
    x = data(n, d, seed, kind)
    h = ora.HNSWIndex(metric, m, ef, 16, d, seed=seed)
    for i in range(n):
        h.insert(i * 3 + 1, x[i])            # ids are not slot numbers
    return h, x


def key_scale(metric, keys, q=None, rows=None):
    """Magnitude the 1e-5 bar is relative to: the distance itself for L2; 1 for a cosine distance (a difference of
    O(1) terms); |q||x| for a dot product, whose value cancels towards 0 while its rounding error does not."""
    if metric == 2: return np.maximum(np.abs(np.asarray(keys, np.float64)), 1e-6)
    if metric == 0: return np.ones(len(keys))
    return np.maximum(np.linalg.norm(q.astype(np.float64)) * np.linalg.norm(rows.astype(np.float64), axis=1), 1e-6)


def compare_results(ref, refn, got, gotn, exact, metric=2, q=None, x=None, row_of=lambda i: (int(i) - 1) // 3):
    """exact: bit-identical records.  Otherwise every key within REL of the oracle's (relative to key_scale) and an
    id may differ only inside a run of keys within TIE of each other; returns the number of ids beyond that."""
    assert np.array_equal(refn, gotn)
    bad = 0
    for qi, (r, g, n) in enumerate(zip(ref, got, refn)):
        r, g = r[:n], g[:n]
        if exact:
            assert np.array_equal(r["id"], g["id"]) and np.array_equal(r["raw_score"].view(np.uint32), g["raw_score"].view(np.uint32)) \
                and np.array_equal(r["sort_key"].view(np.uint32), g["sort_key"].view(np.uint32))
            continue
        rows = x[[row_of(i) for i in r["id"]]] if metric == 1 else None
        scale = key_scale(metric, r["sort_key"], None if q is None else q[qi], rows)
        err = np.abs(r["sort_key"].astype(np.float64) - g["sort_key"].astype(np.float64))
        assert np.all(err <= REL * scale), (qi, (err / scale).max())
        for j in np.nonzero(r["id"] != g["id"])[0]:
            # an id may only move within a run of near-equal keys
            k = r["sort_key"][j]
            near = np.abs(r["sort_key"].astype(np.float64) - k) <= TIE * scale[j]
            if g["id"][j] not in r["id"][near]:
                bad += 1
    return bad


@pytest.mark.parametrize("metric", [0, 1, 2])
@pytest.mark.parametrize("dim,kind", [(32, "uniform"), (128, "clustered"), (100, "uniform")])
def test_search_same_graph_strict_exact_fast_close(ora, vdb, metric, dim, kind):
    n, nq, k = 3000, 200, 10
    h, x = build_oracle(ora, metric, n, dim, seed=7 + metric, kind=kind)
    g = h.export_graph()
    q = data(nq, dim, 99, kind)
    ref, refn, rst = h.search_batch(q, k, with_stats=True)
    for arith in (vdb.Arith.Strict, vdb.Arith.Fast):
        # beam_width=1: FAST expands strictly best-first like the reference, so ids must agree
        p = vdb.HNSWIndex(metric, 16, 64, 16, dim, seed=1, arith=arith, beam_width=1)
        p.import_graph(g)
        got, gotn, st = p.search_batch(q, k, with_stats=True)
        bad = compare_results(ref, refn, got, gotn, exact=(arith == vdb.Arith.Strict), metric=metric, q=q, x=x)
        assert bad <= nq * k // 200, f"{bad} ids differ beyond near-ties"
        if arith == vdb.Arith.Strict:
            # the traversal is the same one: identical work counters
            assert np.array_equal(rst["expanded"], st["expanded"])
    # other beam widths on the same graph
    for ef in (16, 128, 300):
        ref, refn = h.search_batch(q[:50], 20, ef_search=ef)
        p = vdb.HNSWIndex(metric, 16, 64, 16, dim, seed=1, arith=vdb.Arith.Strict)
        p.import_graph(g)
        got, gotn = p.search_batch(q[:50], 20, ef_search=ef)
        compare_results(ref, refn, got, gotn, exact=True)


def _recall(ann, ann_n, exact, exact_n, k):
    hits = 0
    for a, an, e, en in zip(ann, ann_n, exact, exact_n):
        hits += len(set(a["id"][:an].tolist()) & set(e["id"][:en].tolist()))
    return hits / (len(ann) * k)


@pytest.mark.parametrize("metric", [0, 1, 2])
@pytest.mark.parametrize("width", [0, 2, 4])
def test_wide_steps_keep_recall_and_distances(ora, vdb, metric, width):
    """FAST with several nodes expanded per step (the default): recall@10 within 0.5 pt of the oracle's at equal ef
    on the same graph, every returned distance within 1e-5 relative of the strict score of that id, lists sorted."""
    n, d, nq, k = 6000, 64, 300, 10
    h, x = build_oracle(ora, metric, n, d, seed=31 + metric, kind="clustered")
    q = data(nq, d, 77, "clustered")
    exact, exact_n = h.exact_knn_batch(q, k)
    p = vdb.HNSWIndex(metric, 16, 64, 16, d, seed=1, arith=vdb.Arith.Fast, beam_width=width)
    p.import_graph(h.export_graph())
    for ef in (32, 64, 200, 320, 512):                            # 320 and 512: register lists of six and eight chunks
        ref, refn = h.search_batch(q, k, ef_search=ef)
        got, gotn, st = p.search_batch(q, k, ef_search=ef, with_stats=True)
        assert np.array_equal(refn, gotn) and not (st["flags"] & 1).any()
        r_ref, r_got = _recall(ref, refn, exact, exact_n, k), _recall(got, gotn, exact, exact_n, k)
        assert r_got >= r_ref - 0.005, (ef, r_ref, r_got)
        pos = {int(i * 3 + 1): i for i in range(n)}
        for qi in range(0, nq, 7):
            ids = got["id"][qi][: gotn[qi]]
            keys = got["sort_key"][qi][: gotn[qi]]
            assert np.all(np.diff(keys) >= 0)
            rows = x[[pos[int(i)] for i in ids]]
            strict = np.array([ora.score(q[qi], r, metric) for r in rows], np.float32)
            scale = {0: np.ones(len(ids)), 1: np.linalg.norm(q[qi]) * np.linalg.norm(rows, axis=1), 2: np.abs(strict)}[metric]
            assert np.all(np.abs(got["raw_score"][qi][: gotn[qi]] - strict) <= REL * np.maximum(scale, 1e-6))


@pytest.mark.parametrize("metric", [0, 1, 2])
def test_single_insert_builds_the_same_graph(ora, vdb, metric):
    """Graph producer parity (hnsw.rs:119-208): same seed, same points, STRICT arithmetic -> identical lists."""
    n, d = 400, 24
    x = data(n, d, 21)
    ho = ora.HNSWIndex(metric, 8, 32, 16, d, seed=5)
    hp = vdb.HNSWIndex(metric, 8, 32, 16, d, seed=5, arith=vdb.Arith.Strict)
    for i in range(n):
        ho.insert(i + 10, x[i]); hp.insert(i + 10, x[i])
    go, gp = ho.export_graph(), hp.export_graph()
    assert go["entry"] == gp["entry"] and go["current_max_level"] == gp["current_max_level"]
    assert np.array_equal(go["levels"], gp["levels"])
    assert np.array_equal(go["vectors"].view(np.uint32), gp["vectors"].view(np.uint32))
    assert len(go["layers"]) == len(gp["layers"])
    for (ow, of, nb), (ow2, of2, nb2) in zip(go["layers"], gp["layers"]):
        assert np.array_equal(ow, ow2) and np.array_equal(of, of2) and np.array_equal(nb, nb2)


def test_single_inserts_patch_the_upper_layers(vdb, monkeypatch):
    """Single inserts bring the packed upper layers on the GPU up to date list by list (new owners, relocated lists,
    fused blocks) instead of re-packing them; VDB_VERIFY_DEVICE makes every sync compare the device copy with the host
    lists.  A point is found by its own vector as soon as its insert returns, and the final graph answers like the
    same graph imported (and therefore packed from scratch) into a fresh index."""
    monkeypatch.setenv("VDB_VERIFY_DEVICE", "1")
    n0, extra, d, k = 4000, 800, 32, 10
    x = data(n0 + extra, d, 77)            # uniform rows: one connected graph, so a walk always reaches its own point
    h = vdb.HNSWIndex(vdb.DistanceMetric.Euclidean, 8, 48, 16, d, seed=9)
    h.insert_batch(np.arange(n0, dtype=np.uint64), x[:n0])
    for i in range(n0, n0 + extra):
        h.insert(i, x[i])
        if i % 3 == 0:
            out = h.search(x[i], 1)
            assert out[0].id == i and out[0].raw_score <= 1e-3, "a point is findable as soon as its insert returns"
    g = h.export_graph()
    assert len(g["ids"]) == n0 + extra and int(g["levels"][n0:].max()) >= 1, "some new points own upper lists"
    owners0 = set(g["layers"][0][0].tolist())
    assert all(s in owners0 for s in range(n0, n0 + extra))
    q = data(300, d, 78)
    got, gotn = h.search_batch(q, k, ef_search=64)
    fresh = vdb.HNSWIndex(vdb.DistanceMetric.Euclidean, 8, 48, 16, d, seed=9); fresh.import_graph(g)
    ref, refn = fresh.search_batch(q, k, ef_search=64)
    truth, _ = fresh.exact_knn_batch(q, k)
    rec = lambda o: np.mean([len(set(o["id"][i].tolist()) & set(truth["id"][i].tolist())) / k for i in range(len(q))])
    assert np.array_equal(got["id"], ref["id"]) and np.array_equal(gotn, refn), (rec(got), rec(ref))


@pytest.mark.parametrize("metric", [0, 1, 2])
def test_exact_knn_bit_exact(ora, vdb, metric):
    n, d, nq, k = 5000, 48, 64, 10
    h, x = build_oracle(ora, metric, 200, d, seed=3)          # small graph; exact k-NN ignores it
    g = h.export_graph()
    # grow the point set without graph edges: exact k-NN scans rows only
    xs = data(n, d, 8)
    if metric == 0:
        xs = xs / np.linalg.norm(xs, axis=1, keepdims=True).astype(np.float32)
    g2 = dict(g); g2["ids"] = np.arange(n, dtype=np.uint64) * 2 + 5; g2["levels"] = np.zeros(n, np.uint32)
    g2["vectors"] = xs; g2["deleted"] = np.zeros(n, np.uint8); g2["deleted"][::17] = 1
    g2["layers"] = []; g2["entry"] = 5 + 2; g2["current_max_level"] = 0
    ho = ora.HNSWIndex(metric, 16, 64, 16, d); ho.import_graph(g2)
    hp = vdb.HNSWIndex(metric, 16, 64, 16, d); hp.import_graph(g2)
    q = data(nq, d, 4)
    ref, refn = ho.exact_knn_batch(q, k)
    got, gotn = hp.exact_knn_batch(q, k)
    assert np.array_equal(refn, gotn)
    assert np.array_equal(ref["id"], got["id"])
    assert np.array_equal(ref["raw_score"].view(np.uint32), got["raw_score"].view(np.uint32))
    deleted_ids = set(g2["ids"][::17].tolist())
    assert not (set(got["id"].ravel().tolist()) & deleted_ids)


@pytest.mark.parametrize("metric", [0, 1, 2])
@pytest.mark.parametrize("dim", [3, 32, 100, 128, 200, 260, 515, 768, 960])
def test_exact_knn_matrix_core_pass_matches_brute_force(ora, vdb, metric, dim):
    """FAST exact k-NN (MFMA candidate pass + strict re-rank) against the oracle's brute force: same ids, same bits,
    over several chunks and query blocks, with tombstones and repeated rows; STRICT index gives the same answer."""
    n, nq, k = (20000, 300, 10) if dim <= 256 else (9000, 200, 10)      # rows above 256 floats take the chunked kernel
    kind = "clustered" if dim >= 100 else "uniform"
    xs = data(n, dim, 50 + dim, kind)
    xs[1000:1008] = xs[7]                                     # repeated rows: exact ties, ordered by id
    if metric == 0:
        xs = xs / np.maximum(np.linalg.norm(xs, axis=1, keepdims=True), 1e-20).astype(np.float32)
    ids = (np.arange(n, dtype=np.uint64)[::-1] * 3 + 1).copy()  # ids descend with the slot
    dele = np.zeros(n, np.uint8); dele[::13] = 1
    g = dict(ids=ids, levels=np.zeros(n, np.uint32), vectors=xs, deleted=dele, layers=[], entry=int(ids[1]), current_max_level=0)
    q = data(nq, dim, 51, kind); q[0] = xs[7]
    ho = ora.HNSWIndex(metric, 16, 64, 16, dim); ho.import_graph(g)
    ref, refn = ho.exact_knn_batch(q, k)
    for arith in (vdb.Arith.Fast, vdb.Arith.Strict):
        hp = vdb.HNSWIndex(metric, 16, 64, 16, dim, arith=arith); hp.import_graph(g)
        got, gotn = hp.exact_knn_batch(q, k)
        assert np.array_equal(refn, gotn)
        assert np.array_equal(ref["id"], got["id"]), (arith, np.argwhere(ref["id"] != got["id"])[:5])
        assert np.array_equal(ref["raw_score"].view(np.uint32), got["raw_score"].view(np.uint32))
        assert np.array_equal(ref["sort_key"].view(np.uint32), got["sort_key"].view(np.uint32))


@pytest.mark.parametrize("dim", [128, 768])
def test_exact_knn_far_from_the_origin_matches_brute_force(ora, vdb, dim):
    """Euclidean rows that share a large common component (every coordinate near 1000, spread 1): |x|^2 is 10^6 times a
    neighbour's squared distance, so an error relative to the products q_i x_i swamps the gaps between neighbours
    (without the centring this test fails: VDB_KNN_NO_CENTRE=1).  The matrix-core pass works on rows and queries
    minus a common centre; ids and bits equal the oracle's brute force."""
    n, nq, k = (30000, 200, 10) if dim <= 256 else (9000, 100, 10)
    rng = np.random.default_rng(900 + dim)
    xs = (1000.0 + rng.standard_normal((n, dim))).astype(np.float32)
    q = (1000.0 + rng.standard_normal((nq, dim))).astype(np.float32)
    ids = np.arange(n, dtype=np.uint64) * 2 + 5
    g = dict(ids=ids, levels=np.zeros(n, np.uint32), vectors=xs, deleted=np.zeros(n, np.uint8), layers=[], entry=int(ids[0]), current_max_level=0)
    ho = ora.HNSWIndex(2, 16, 64, 16, dim); ho.import_graph(g)
    ref, refn = ho.exact_knn_batch(q, k)
    hp = vdb.HNSWIndex(2, 16, 64, 16, dim, arith=vdb.Arith.Fast); hp.import_graph(g)
    got, gotn = hp.exact_knn_batch(q, k)
    assert np.array_equal(refn, gotn)
    assert np.array_equal(ref["id"], got["id"]), np.argwhere(ref["id"] != got["id"])[:5]
    assert np.array_equal(ref["sort_key"].view(np.uint32), got["sort_key"].view(np.uint32))


@pytest.mark.parametrize("metric", [0, 1, 2])
@pytest.mark.parametrize("case", ["block-of-40", "period-97"])
def test_exact_knn_many_exact_duplicates_rank_by_id(ora, vdb, metric, case):
    """More exact duplicates around the k-th place than the matrix-core pass keeps spare candidates, with ids that
    DESCEND with the slot: the answer is still the (sort_key, id) order of the oracle's brute force, bit for bit.
    "period-97" is the data of tests/in_place.rs:17-19 (every vector occurs about 51 times)."""
    k, nq = 10, 64
    if case == "block-of-40":
        n, dim = 20000, 64
        xs = data(n, dim, 300 + metric, "clustered")
        xs[7] *= 3                                             # the repeated row also wins every dot product
        xs[1000:1040] = xs[7]                                  # 41 copies of one row
        xs[15000:15040] = xs[7]                                # and 40 more far away
        q = data(nq, dim, 301, "clustered"); q[0] = xs[7]; q[1] = xs[7] * 1.01
    else:
        n, dim = 5000, 32
        xs = np.array([[((i + j * 7919) % 97) / 10.0 for j in range(dim)] for i in range(n)], np.float32)
        q = xs[np.arange(nq) * 3 + 1].copy()
    if metric == 0:
        xs = xs / np.maximum(np.linalg.norm(xs, axis=1, keepdims=True), 1e-20).astype(np.float32)
    ids = (np.arange(n, dtype=np.uint64)[::-1] * 3 + 1).copy()  # ids descend with the slot
    g = dict(ids=ids, levels=np.zeros(n, np.uint32), vectors=xs, deleted=np.zeros(n, np.uint8), layers=[], entry=int(ids[1]), current_max_level=0)
    ho = ora.HNSWIndex(metric, 16, 64, 16, dim); ho.import_graph(g)
    ref, refn = ho.exact_knn_batch(q, k)
    assert len(set(ref["sort_key"][0].tolist())) == 1, "the case is meant to fill the whole top-k with one tie class"
    hp = vdb.HNSWIndex(metric, 16, 64, 16, dim, arith=vdb.Arith.Fast); hp.import_graph(g)
    got, gotn = hp.exact_knn_batch(q, k)
    assert np.array_equal(refn, gotn)
    assert np.array_equal(ref["id"], got["id"]), np.argwhere(ref["id"] != got["id"])[:5]
    assert np.array_equal(ref["sort_key"].view(np.uint32), got["sort_key"].view(np.uint32))


@pytest.mark.parametrize("name,metric,n,dim,nq,ef", [
    ("C1", 0, 10000, 128, 1000, 64),          # 10k x 128 cosine, uniform(-1, 1), M=16, ef=64
    ("C3-reduced", 0, 6000, 960, 200, 64),    # 960-d cosine rows (3840 B), clustered
    ("wide-rows", 2, 4000, 200, 200, 100),    # 200-d L2: one row per wavefront pass
])
def test_reference_configurations(ora, vdb, name, metric, n, dim, nq, ef):
    """The smaller benchmark configurations as parity cases: STRICT search bit-identical to the oracle on the
    oracle-built graph, FAST (one node per step) ids equal outside near-ties, FAST default recall within 0.5 pt."""
    k = 10
    kind = "uniform" if name == "C1" else "clustered"
    h, x = build_oracle(ora, metric, n, dim, seed=0xC1, kind=kind)
    g = h.export_graph()
    q = data(nq, dim, 0xC1 + 1, kind)
    ref, refn, rst = h.search_batch(q, k, ef_search=ef, with_stats=True)
    exact, exact_n = h.exact_knn_batch(q, k)
    ps = vdb.HNSWIndex(metric, 16, 64, 16, dim, arith=vdb.Arith.Strict); ps.import_graph(g)
    got, gotn, st = ps.search_batch(q, k, ef_search=ef, with_stats=True)
    compare_results(ref, refn, got, gotn, exact=True)
    assert np.array_equal(rst["expanded"], st["expanded"])
    e2, e2n = ps.exact_knn_batch(q, k)
    assert np.array_equal(exact["id"], e2["id"]) and np.array_equal(exact["raw_score"].view(np.uint32), e2["raw_score"].view(np.uint32))
    p1 = vdb.HNSWIndex(metric, 16, 64, 16, dim, arith=vdb.Arith.Fast, beam_width=1); p1.import_graph(g)
    got, gotn = p1.search_batch(q, k, ef_search=ef)
    assert compare_results(ref, refn, got, gotn, exact=False, metric=metric, q=q, x=x) <= nq * k // 200
    pw = vdb.HNSWIndex(metric, 16, 64, 16, dim, arith=vdb.Arith.Fast); pw.import_graph(g)
    got, gotn, st = pw.search_batch(q, k, ef_search=ef, with_stats=True)
    assert not (st["flags"] & 1).any()
    assert _recall(got, gotn, exact, exact_n, k) >= _recall(ref, refn, exact, exact_n, k) - 0.005
    e3, e3n = pw.exact_knn_batch(q, k)
    assert np.array_equal(exact["id"], e3["id"]) and np.array_equal(exact["raw_score"].view(np.uint32), e3["raw_score"].view(np.uint32))


def test_high_degree_nodes_deleted_rows_and_long_lists(ora, vdb):
    """Register-resident beam off the beaten path: nodes with more than 62 neighbours (overflow chunks of the
    adjacency block), tombstones inside the wide step, ef > 256 (LDS list) — all against the oracle."""
    metric, n, d, nq, k = 2, 5000, 16, 200, 10
    rng = np.random.default_rng(3)
    x = rng.normal(0, 1, size=(n, d)).astype(np.float32)
    x[:40] *= 0.05                                           # a tight core: every later point links to it
    h = ora.HNSWIndex(metric, 16, 64, 16, d, seed=9)
    for i in range(n):
        h.insert(i * 3 + 1, x[i])
    g = h.export_graph()
    ow, of, nb = g["layers"][0]
    assert (np.diff(of.astype(np.int64)) > 62).sum() >= 5, "the data set is meant to produce high-degree nodes"
    q = rng.normal(0, 1, size=(nq, d)).astype(np.float32)
    kill = [int(v) for v in rng.choice(g["ids"], size=600, replace=False)]
    ps = vdb.HNSWIndex(metric, 16, 64, 16, d, arith=vdb.Arith.Strict); ps.import_graph(g)
    p1 = vdb.HNSWIndex(metric, 16, 64, 16, d, beam_width=1); p1.import_graph(g)
    pw = vdb.HNSWIndex(metric, 16, 64, 16, d); pw.import_graph(g)
    for phase in ("fresh", "deleted"):
        if phase == "deleted":
            for pid in kill:
                h.mark_deleted(pid); ps.mark_deleted(pid); p1.mark_deleted(pid); pw.mark_deleted(pid)
        exact, exact_n = h.exact_knn_batch(q, k)
        for ef in (64, 200, 400):
            ref, refn = h.search_batch(q, k, ef_search=ef)
            got, gotn = ps.search_batch(q, k, ef_search=ef)
            compare_results(ref, refn, got, gotn, exact=True)
            got, gotn = p1.search_batch(q, k, ef_search=ef)
            assert compare_results(ref, refn, got, gotn, exact=False, metric=metric, q=q, x=x) <= nq * k // 200
            got, gotn = pw.search_batch(q, k, ef_search=ef)
            assert np.array_equal(refn, gotn)
            assert _recall(got, gotn, exact, exact_n, k) >= _recall(ref, refn, exact, exact_n, k) - 0.005
            if phase == "deleted":
                assert not (set(got["id"].ravel().tolist()) & set(kill))


def test_a_small_visited_table_changes_the_work_not_the_results(ora, vdb, monkeypatch):
    """The visited table of the register beam forgets ids when a bucket is full; a forgotten id is scored again and
    then rejected by the bound or found in the list.  A table far too small for the walk (2048 tags, ef = 256) must
    return exactly what the largest table returns, with more rows scored and no flag raised."""
    metric, n, d, nq, k = 2, 20000, 32, 300, 10
    h, x = build_oracle(ora, metric, n, d, seed=44, kind="clustered")
    g = h.export_graph()
    q = data(nq, d, 45, "clustered")
    exact, exact_n = h.exact_knn_batch(q, k)
    ref, refn = h.search_batch(q, k, ef_search=256)
    p = vdb.HNSWIndex(metric, 16, 64, 16, d); p.import_graph(g)
    monkeypatch.setenv("VDB_TAG_LOG2", "15")
    big, bign, st_big = p.search_batch(q, k, ef_search=256, with_stats=True)
    monkeypatch.setenv("VDB_TAG_LOG2", "11")
    got, gotn, st = p.search_batch(q, k, ef_search=256, with_stats=True)
    monkeypatch.setenv("VDB_TAG_LOG2", "7")                   # 128 tags: nearly every insert replaces an id
    tiny, tinyn, st_tiny = p.search_batch(q, k, ef_search=256, with_stats=True)
    monkeypatch.delenv("VDB_TAG_LOG2")
    dflt, dfltn, st_d = p.search_batch(q, k, ef_search=256, with_stats=True)
    assert not (st["flags"] & 1).any() and not (st_big["flags"] & 1).any() and not (st_tiny["flags"] & 1).any()
    for a, an in ((got, gotn), (dflt, dfltn), (tiny, tinyn)):
        assert np.array_equal(bign, an) and np.array_equal(big["id"], a["id"])
        assert np.array_equal(big["sort_key"].view(np.uint32), a["sort_key"].view(np.uint32))
    assert np.array_equal(st_big["expanded"], st["expanded"]) and np.array_equal(st_big["expanded"], st_tiny["expanded"])   # same walk
    assert (st["rows_scored"] >= st_big["rows_scored"]).all() and (st["rows_scored"] > st_big["rows_scored"]).any(), \
        "the case is meant to make the small table forget"
    assert st_tiny["rows_scored"].mean() > 1.5 * st_big["rows_scored"].mean()
    assert np.array_equal(refn, gotn)
    assert _recall(got, gotn, exact, exact_n, k) >= _recall(ref, refn, exact, exact_n, k) - 0.005


@pytest.mark.parametrize("metric", [0, 1])
@pytest.mark.parametrize("dim", [128, 768])
def test_exact_knn_with_a_large_shared_component_matches_brute_force(ora, vdb, monkeypatch, metric, dim):
    """Cosine / dot rows with one large common component (all-positive embeddings, small angular spread): neighbours
    differ by less than the candidate pass's bf16-split error, which no centring removes for these metrics.  The
    re-rank compares the strict k-th key with the bounds the candidate pass dropped rows at and, where a dropped row
    cannot be excluded, scores every row for that query: the oracle's brute force bit for bit.  Without that check
    (VDB_KNN_NO_VERIFY=1) this data loses neighbours — the case is meant to need it."""
    n, nq, k = (30000, 200, 10) if dim <= 256 else (9000, 100, 10)
    rng = np.random.default_rng(700 + dim + metric)
    xs = (100.0 + rng.standard_normal((n, dim)) * 0.05).astype(np.float32)
    q = (100.0 + rng.standard_normal((nq, dim)) * 0.05).astype(np.float32)
    xs = xs / np.linalg.norm(xs, axis=1, keepdims=True).astype(np.float32)     # unit rows for both metrics (embedding-like)
    ids = np.arange(n, dtype=np.uint64) * 2 + 5
    g = dict(ids=ids, levels=np.zeros(n, np.uint32), vectors=xs, deleted=np.zeros(n, np.uint8), layers=[], entry=int(ids[0]), current_max_level=0)
    ho = ora.HNSWIndex(metric, 16, 64, 16, dim); ho.import_graph(g)
    ref, refn = ho.exact_knn_batch(q, k)
    hp = vdb.HNSWIndex(metric, 16, 64, 16, dim, arith=vdb.Arith.Fast); hp.import_graph(g)
    got, gotn = hp.exact_knn_batch(q, k)
    assert np.array_equal(refn, gotn)
    assert np.array_equal(ref["id"], got["id"]), np.argwhere(ref["id"] != got["id"])[:5]
    assert np.array_equal(ref["sort_key"].view(np.uint32), got["sort_key"].view(np.uint32))
    monkeypatch.setenv("VDB_KNN_NO_VERIFY", "1")
    raw, _ = hp.exact_knn_batch(q, k)
    assert not np.array_equal(ref["id"], raw["id"]), "the unverified candidate pass was expected to lose neighbours here"


def test_exact_knn_ties_break_by_id(ora, vdb):
    d = 8
    base = np.zeros((40, d), np.float32); base[:, 0] = np.repeat(np.arange(10), 4)     # 4-way exact ties
    ids = np.array([100 - i for i in range(40)], dtype=np.uint64)                      # descending ids
    g = dict(ids=ids, levels=np.zeros(40, np.uint32), vectors=base, deleted=np.zeros(40, np.uint8), layers=[], entry=int(ids[0]), current_max_level=0)
    ho = ora.HNSWIndex(2, 16, 64, 16, d); ho.import_graph(g)
    hp = vdb.HNSWIndex(2, 16, 64, 16, d); hp.import_graph(g)
    q = np.zeros((1, d), np.float32)
    ref, _ = ho.exact_knn_batch(q, 10); got, _ = hp.exact_knn_batch(q, 10)
    assert np.array_equal(ref["id"], got["id"])
    assert list(got["id"][0][:4]) == sorted(got["id"][0][:4])


def test_deleted_points_and_fallback_entry(ora, vdb):
    metric, d = 2, 16
    h, x = build_oracle(ora, metric, 800, d, seed=12)
    g = h.export_graph()
    p = vdb.HNSWIndex(metric, 16, 64, 16, d, arith=vdb.Arith.Strict); p.import_graph(g)
    rng = np.random.default_rng(0)
    kill = rng.choice(g["ids"], size=200, replace=False)
    kill = np.concatenate([kill, [h.get_entry_point()]])          # tombstone the entry too (hnsw.rs:114-116)
    for pid in kill:
        h.mark_deleted(int(pid)); p.mark_deleted(int(pid))
    assert h.get_entry_point() == p.get_entry_point()
    q = data(100, d, 77)
    ref, refn = h.search_batch(q, 10); got, gotn = p.search_batch(q, 10)
    compare_results(ref, refn, got, gotn, exact=True)
    assert not (set(got["id"].ravel().tolist()) & set(int(v) for v in kill))


def test_topk_merge_matches_oracle(ora, vdb):
    import ctypes as C
    torch = pytest.importorskip("torch")
    rng = np.random.default_rng(3)
    S, Q, K = 4, 37, 10
    lists = np.zeros((S, Q, K), dtype=vdb.SCORED_DTYPE); counts = rng.integers(0, K + 1, size=(S, Q)).astype(np.uint32)
    for s in range(S):
        for q in range(Q):
            keys = np.sort(rng.integers(0, 30, size=K).astype(np.float32))      # many cross-shard ties
            lists[s, q]["sort_key"] = keys; lists[s, q]["raw_score"] = keys
            lists[s, q]["id"] = rng.permutation(1000)[:K] + s * 1000
            o = np.lexsort((lists[s, q]["id"], lists[s, q]["sort_key"])); lists[s, q] = lists[s, q][o]
    ref = np.zeros((Q, K), dtype=vdb.SCORED_DTYPE); refn = np.zeros(Q, np.uint32)
    ora.check(ora.fn.topk_merge(lists.ctypes.data, counts.ctypes.data, S, Q, K, ref.ctypes.data, refn.ctypes.data))
    dl = torch.from_numpy(lists.view(np.uint8).reshape(-1)).cuda(); dc = torch.from_numpy(counts.view(np.int32).reshape(-1)).cuda()
    do = torch.zeros(Q * K * 16, dtype=torch.uint8, device="cuda"); dn = torch.zeros(Q, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    vdb.check(vdb.fn.topk_merge_dev(dl.data_ptr(), dc.data_ptr(), S, Q, K, do.data_ptr(), dn.data_ptr(), None))
    torch.cuda.synchronize()
    got = do.cpu().numpy().view(vdb.SCORED_DTYPE).reshape(Q, K); gotn = dn.cpu().numpy().view(np.uint32)
    assert np.array_equal(refn, gotn)
    for q in range(Q):
        assert np.array_equal(ref[q][: refn[q]], got[q][: gotn[q]])


def _payload_segment(api, metric, x, mode, arith=None):
    kw = dict(seed=9, filter_mode=mode)
    if arith is not None: kw["arith"] = arith
    seg = api.Segment(api.HNSWIndex(metric, 8, 32, 16, x.shape[1], **kw))
    for i in range(len(x)):
        p = api.Payload(); p.set("cat", api.PayloadValue.Int(i % 20)); p.set("half", api.PayloadValue.Int(i % 2))
        seg.insert(x[i], p)
    return seg


@pytest.mark.parametrize("metric", [0, 1, 2])
def test_masked_filter_mode_walks_the_graph_like_the_oracle(ora, vdb, monkeypatch, metric):
    """Masked mode (the beam runs over every row, only passing rows enter the result list): same segment built by
    single inserts on both sides, STRICT arithmetic -> identical results."""
    n, d, k = 700, 16, 10
    x = data(n, d, 81 + metric); q = data(25, d, 82)
    so = _payload_segment(ora, metric, x, ora.FilterMode.Masked); so.set_filter_mode(ora.FilterMode.Masked)
    sp = _payload_segment(vdb, metric, x, vdb.FilterMode.Masked, arith=vdb.Arith.Strict)
    monkeypatch.setenv("VDB_MASKED_SCAN", "0")               # keep the graph walk even for narrow filters
    for make in (lambda a: a.Filter.Match("half", a.PayloadValue.Int(1)),
                 lambda a: a.Filter.Match("cat", a.PayloadValue.Int(3)),
                 lambda a: a.Filter.And([a.Filter.Match("half", a.PayloadValue.Int(0)), a.Filter.Compare("cat", a.ScalarComparisonOp.Lt, a.PayloadValue.Int(9))])):
        for qi in range(len(q)):
            ref = so.in_place_filtered_search(q[qi], k, make(ora)); got = sp.in_place_filtered_search(q[qi], k, make(vdb))
            assert [r.id for r in ref] == [g.id for g in got]
            assert [np.float32(r.sort_key).view(np.uint32) for r in ref] == [np.float32(g.sort_key).view(np.uint32) for g in got]


@pytest.mark.parametrize("metric", [0, 1, 2])
@pytest.mark.parametrize("arith", ["strict", "fast"])
def test_masked_mode_scans_the_passing_rows_of_a_narrow_filter(ora, vdb, metric, arith):
    """At most 1/8 of the rows pass: Masked mode returns the exact nearest passing rows (checked against the oracle's
    brute force over an index that holds only the passing rows), tombstones excluded."""
    n, d, k, nq = 900, 24, 10, 40
    x = data(n, d, 91 + metric); q = data(nq, d, 92)
    ar = vdb.Arith.Strict if arith == "strict" else vdb.Arith.Fast
    seg = vdb.Segment(vdb.HNSWIndex(metric, 8, 32, 16, d, seed=3, filter_mode=vdb.FilterMode.Masked, arith=ar))
    pay = []
    for i in range(n):
        p = vdb.Payload(); p.set("cat", vdb.PayloadValue.Int(i % 20)); pay.append(p)
    ids = seg.insert_batch(x, pay)
    passing = [i for i in range(n) if i % 20 == 3]
    seg.delete(int(ids[passing[0]])); seg.delete(int(ids[passing[5]])); seg.delete(int(ids[7]))
    live = [i for i in passing if i not in (passing[0], passing[5])]
    ho = ora.HNSWIndex(metric, 8, 32, 16, d, seed=3)
    for i in live: ho.insert(int(ids[i]), x[i])
    ref, refn = ho.exact_knn_batch(q, k)
    f = vdb.Filter.Match("cat", vdb.PayloadValue.Int(3))
    got, gotn = seg.search_batch(q, k, filter=f)
    bad = compare_results(ref, refn, got, gotn, exact=(arith == "strict"), metric=metric, q=q, x=x, row_of=lambda i: int(i) - 1)
    assert bad == 0
    one = seg.search_with_filter(q[0], k, f)
    assert [r.id for r in one] == got["id"][0][: gotn[0]].tolist()
    wide = seg.in_place_filtered_search(q[1], 30, f)                      # more than the MFMA pass keeps per chunk
    ref30, n30 = ho.exact_knn_batch(q[1:2], 30)
    assert [r.id for r in wide] == ref30["id"][0][: n30[0]].tolist() or arith == "fast"
    assert len(wide) == n30[0]


def test_literal_filtered_search_reports_a_candidate_queue_that_is_too_small(vdb, monkeypatch):
    """The literal walk keeps every scored row in its candidate queue; a queue that cannot hold them is an error
    that names the way out, and with room for one entry per row the same call succeeds."""
    n, d = 600, 8
    x = data(n, d, 101)
    seg = _payload_segment(vdb, 2, x, vdb.FilterMode.Literal)
    f = vdb.Filter.Match("half", vdb.PayloadValue.Int(1))
    full = seg.in_place_filtered_search(x[5], 10, f)
    assert len(full) == 10
    monkeypatch.setenv("VDB_LITERAL_CAND_CAP", "48")
    with pytest.raises(vdb.DBError, match="VDB_FILTER_MASKED"):
        seg.in_place_filtered_search(x[5], 10, f)


@pytest.mark.parametrize("metric", [1, 2])
def test_batch_insert_adds_the_filter_aware_edges(ora, vdb, metric):
    """Segment batch inserts add the extra edges of hnsw.rs:210-325 with two launches per batch: every point gets
    its self-edge (hnsw.rs:318) and, for its first key, edges to the m nearest points that share the value (posting
    lists of at most 101 points are not sampled, so the set is exact); filtered search over the result finds the
    passing neighbours."""
    n, d, m = 3000, 16, 8
    x = data(n, d, 111 + metric)
    seg = vdb.Segment(vdb.HNSWIndex(metric, m, 32, 16, d, seed=4, arith=vdb.Arith.Strict, filter_mode=vdb.FilterMode.Masked))
    pay = []
    for i in range(n):
        p = vdb.Payload(); p.set("cat", vdb.PayloadValue.Int(i % 60)); pay.append(p)
    ids = seg.insert_batch(x, pay)
    h = seg.hnsw()
    row = {int(ids[i]): i for i in range(n)}
    for i in range(0, n, 37):
        nb = h.layer_neighbors(0, int(ids[i]))
        assert int(ids[i]) in nb                                           # the self-edge
        same = [j for j in range(n) if j % 60 == i % 60 and j != i]
        sc = np.array([ora.score(x[i], x[j], metric) for j in same], np.float32)
        order = np.argsort(-sc if metric == 1 else sc, kind="stable")[:m]
        want = {int(ids[same[k]]) for k in order}
        assert want <= set(nb), (i, sorted(want - set(nb)))
        assert sum(1 for t in nb if t != int(ids[i]) and row[t] % 60 == i % 60) >= m
    # the edges are symmetric (add_bidirectional_edge) and searchable
    q = data(20, d, 113)
    f = vdb.Filter.Match("cat", vdb.PayloadValue.Int(7))
    got, gotn = seg.search_batch(q, 5, filter=f)
    assert (gotn == 5).all() and all(row[int(t)] % 60 == 7 for t in got["id"].ravel())


@pytest.mark.parametrize("metric", [0, 1, 2])
@pytest.mark.parametrize("dim,k", [(64, 13), (128, 25), (768, 28), (96, 40)])
def test_exact_knn_with_larger_top_k(ora, vdb, metric, dim, k):
    """top_k up to 28 keeps 32 candidates per (query, chunk) in the matrix-core pass; above that every pair is scored
    in strict arithmetic.  Both must equal the oracle's brute force bit for bit."""
    n, nq = 12000, 150
    xs = data(n, dim, 70 + dim, "clustered")
    if metric == 0:
        xs = xs / np.maximum(np.linalg.norm(xs, axis=1, keepdims=True), 1e-20).astype(np.float32)
    ids = np.arange(n, dtype=np.uint64) * 5 + 2
    dele = np.zeros(n, np.uint8); dele[::7] = 1
    g = dict(ids=ids, levels=np.zeros(n, np.uint32), vectors=xs, deleted=dele, layers=[], entry=int(ids[1]), current_max_level=0)
    q = data(nq, dim, 71, "clustered")
    ho = ora.HNSWIndex(metric, 16, 64, 16, dim); ho.import_graph(g)
    ref, refn = ho.exact_knn_batch(q, k)
    hp = vdb.HNSWIndex(metric, 16, 64, 16, dim, arith=vdb.Arith.Fast); hp.import_graph(g)
    got, gotn = hp.exact_knn_batch(q, k)
    assert np.array_equal(refn, gotn) and np.array_equal(ref["id"], got["id"])
    assert np.array_equal(ref["sort_key"].view(np.uint32), got["sort_key"].view(np.uint32))


def test_incremental_device_copy_equals_a_fresh_pack(vdb, monkeypatch):
    """Batched and single inserts keep the device copy of layer 0 current with block patches for new and re-linked
    lists and single appended entries for reverse edges.  With VDB_VERIFY_DEVICE=1 every sync compares all blocks
    with a fresh pack of the host lists; and an index that imports the exported graph (one full pack) must answer
    every query exactly like the incrementally maintained one."""
    monkeypatch.setenv("VDB_VERIFY_DEVICE", "1")
    n, d, k = 24000, 24, 10
    x = data(n + 300, d, 121, "clustered")
    a = vdb.HNSWIndex(2, 16, 64, 16, d, seed=12)
    for lo in range(0, n, 5000):
        a.insert_batch(np.arange(lo, min(n, lo + 5000), dtype=np.uint64), x[lo:min(n, lo + 5000)])
        a.sync_device()
    for i in range(n, n + 300):                                  # single inserts on top
        a.insert(i, x[i])
        if i % 50 == 0: a.sync_device()
    a.add_bidirectional_edge(0, 5, 17); a.add_bidirectional_edge(0, 5, 17)    # repeated edge: the list is re-packed as a whole
    a.mark_deleted(40)
    a.sync_device()
    info = a.device_info()
    b = vdb.HNSWIndex(2, 16, 64, 16, d, seed=12); b.import_graph(a.export_graph()); b.sync_device()
    assert b.device_info()["l0_edges"] == info["l0_edges"]
    q = data(400, d, 122, "clustered")
    for ef in (32, 150):
        ra, na, sa = a.search_batch(q, k, ef_search=ef, with_stats=True)
        rb, nb, sb = b.search_batch(q, k, ef_search=ef, with_stats=True)
        assert np.array_equal(na, nb) and np.array_equal(ra["id"], rb["id"]) and np.array_equal(sa["expanded"], sb["expanded"])
        assert np.array_equal(ra["sort_key"].view(np.uint32), rb["sort_key"].view(np.uint32))


@pytest.mark.parametrize("metric", [0, 1, 2])
def test_masked_walk_on_the_register_beam(ora, vdb, monkeypatch, metric):
    """FAST arithmetic, wide filter: the masked walk runs on the register-resident beam with the passing rows in a
    register list.  Strictly best-first (beam_width = 1) it must return what the LDS-list kernel returns on the same
    segment (same arithmetic, same order of expansion); at the default width recall against the brute force over
    the passing rows is no worse; a tiny visited table changes nothing; every result passes and no id repeats."""
    monkeypatch.setenv("VDB_MASKED_SCAN", "0")                  # keep the walk although half of the rows pass
    n, d, k, nq = 6000, 32, 10, 300
    x = data(n, d, 131 + metric, "clustered"); q = data(nq, d, 132, "clustered")
    if metric != 2:
        x /= np.linalg.norm(x, axis=1, keepdims=True); q /= np.linalg.norm(q, axis=1, keepdims=True)
    pay = []
    for i in range(n):
        p = vdb.Payload(); p.set("half", vdb.PayloadValue.Int(i % 2)); pay.append(p)
    f = vdb.Filter.Match("half", vdb.PayloadValue.Int(1))

    def segment(width):
        seg = vdb.Segment(vdb.HNSWIndex(metric, 16, 64, 16, d, seed=9, arith=vdb.Arith.Fast, filter_mode=vdb.FilterMode.Masked, beam_width=width))
        ids = seg.insert_batch(x, pay)
        return seg, ids
    s1, ids = segment(1)
    passing = [i for i in range(n) if i % 2 == 1]
    hb = ora.HNSWIndex(metric, 16, 64, 16, d, seed=9)
    for i in passing: hb.insert(int(ids[i]), x[i])
    exact, exact_n = hb.exact_knn_batch(q, k)
    reg, regn = s1.search_batch(q, k, ef_search=96, filter=f)
    monkeypatch.setenv("VDB_FORCE_GENERIC", "1")
    gen, genn = s1.search_batch(q, k, ef_search=96, filter=f)
    monkeypatch.delenv("VDB_FORCE_GENERIC")
    assert compare_results(gen, genn, reg, regn, exact=False, metric=metric, q=q, x=x, row_of=lambda i: int(i) - 1) <= nq * k // 200
    s4, _ = segment(0)
    wide, widen = s4.search_batch(q, k, ef_search=96, filter=f)
    monkeypatch.setenv("VDB_TAG_LOG2", "7")
    tiny, tinyn = s4.search_batch(q, k, ef_search=96, filter=f)
    monkeypatch.delenv("VDB_TAG_LOG2")
    assert np.array_equal(widen, tinyn) and np.array_equal(wide["id"], tiny["id"]) and np.array_equal(wide["sort_key"].view(np.uint32), tiny["sort_key"].view(np.uint32))
    assert _recall(wide, widen, exact, exact_n, k) >= _recall(gen, genn, exact, exact_n, k) - 0.005
    odd = {int(ids[i]) for i in passing}
    for out, cnt in ((reg, regn), (wide, widen)):
        assert (cnt == k).all()
        for i in range(nq):
            row = out["id"][i][: cnt[i]].tolist()
            assert len(set(row)) == len(row) and set(row) <= odd
            assert np.all(np.diff(out["sort_key"][i][: cnt[i]]) >= 0)


def test_literal_filtered_search_beyond_the_old_queue_size_matches_the_oracle(ora, vdb):
    """hnsw.rs:639-689 never prunes its candidate heap; the walk visits every reachable row once ef passing rows are
    known.  A segment of 18 000 rows (the candidate queue used to stop at 16 384 entries) built point by point on both
    sides: same results as the oracle's literal walk, worst-first order and equal-key order included."""
    n, d, k = 18_000, 8, 10
    x = data(n, d, 171); q = data(6, d, 172)
    so = _payload_segment(ora, 2, x, ora.FilterMode.Literal)
    sp = _payload_segment(vdb, 2, x, vdb.FilterMode.Literal, arith=vdb.Arith.Strict)
    for make in (lambda a: a.Filter.Match("half", a.PayloadValue.Int(1)), lambda a: a.Filter.Match("cat", a.PayloadValue.Int(3))):
        for qi in range(len(q)):
            ref = so.in_place_filtered_search(q[qi], k, make(ora)); got = sp.in_place_filtered_search(q[qi], k, make(vdb))
            assert [r.id for r in ref] == [g.id for g in got]
            assert [np.float32(r.sort_key).view(np.uint32) for r in ref] == [np.float32(g.sort_key).view(np.uint32) for g in got]
    out, cnt = sp.search_batch(q, 5, filter=vdb.Filter.Match("half", vdb.PayloadValue.Int(0)))       # the batched form, several queries per launch
    for qi in range(len(q)):
        one = sp.search_with_filter(q[qi], 5, vdb.Filter.Match("half", vdb.PayloadValue.Int(0)))
        assert [r.id for r in one] == out["id"][qi][: cnt[qi]].tolist()
