"""CPU tests: the plain-C oracle (oracle/slic_oracle.c) against the committed reference fixtures, and
against the unmodified reference itself when oracle/_ref has been built (it is, in the build container)."""
import numpy as np
import pytest

from oracle import oracle as orc
from oracle import ref
from fast_slic_amd.synth import variant
from util import digest, small_case_names, cca_case_names, case_kwargs, describe_mismatch


def test_lab_tables_match_reference_digests(golden_digests):
    g, t, cb = orc.lab_tables()
    assert digest(g.astype(np.int32)) == golden_digests["tables"]["gamma_i32"]
    assert digest(t.astype(np.int32)) == golden_digests["tables"]["lab_tbl_i32"]
    # Cb = round(C * 65536)
    assert cb.tolist() == [28440, 24656, 12442, 13938, 46868, 4730, 1164, 7175, 57202]
    # spot values recorded in SURVEY 8c
    assert g[:4].tolist() == [0, 2, 4, 7] and g[255] == 8192
    assert t[0] == 1130 and t[8192] == 8192


def test_lab_known_answers():
    # live values of the reference for src/cpptest/test_cielab.cpp:30-36's inputs
    rgb = np.array([[139, 91, 30], [111, 197, 143], [255, 255, 255], [255, 255, 0], [255, 0, 255],
                    [0, 255, 255], [30, 57, 184], [0, 0, 0]], np.uint8).reshape(1, 8, 3)
    exp = [(85, 156, 210), (146, 51, 166), (200, 128, 128), (194, 84, 255), (120, 255, 6),
           (182, 31, 99), (61, 204, 0), (0, 128, 128)]
    lab = orc.rgb_to_lab(rgb)
    assert [tuple(int(v) for v in lab[0, i, :3]) for i in range(8)] == exp
    assert (lab[..., 3] == 0).all()


def test_small_cases_all_stages(golden_small):
    for name in small_case_names(golden_small):
        img = golden_small[name + "/image"]
        cl0 = golden_small[name + "/clusters0"]
        K = int(golden_small[name + "/K"])
        kw = case_kwargs(golden_small, name)
        assert orc.initialize_clusters(img, K).tobytes() == cl0.tobytes(), name
        labels, cl, lab, pre = orc.slic_iterate(img, cl0, stages=True, **kw)
        for nm, got, exp in (("lab", lab, golden_small[name + "/lab"]),
                             ("prelabels", pre, golden_small[name + "/prelabels"]),
                             ("labels", labels, golden_small[name + "/labels"])):
            assert np.array_equal(got, exp), describe_mismatch(name + "/" + nm, got, exp)
        assert cl.tobytes() == golden_small[name + "/clusters"].tobytes(), name


def test_connectivity_cases(golden_small):
    for name in cca_case_names(golden_small):
        if name.startswith("kat"):
            K, thr = 10, 0
        else:
            K, thr = int(golden_small[name + "/K"]), int(golden_small[name + "/thr"])
        got = orc.enforce_connectivity(golden_small[name + "/in"], K, thr)
        exp = golden_small[name + "/out"]
        assert np.array_equal(got, exp), describe_mismatch(name, got, exp)


@pytest.mark.parametrize("name", ["C1_480x640_k200", "C2_720x1280_k1600", "C_480x640_k200", "D_480x640_k100"])
def test_baseline_sizes_against_reference_digests(golden_digests, name):
    d = golden_digests[name]
    img = variant(d["variant"], d["H"], d["W"])
    assert digest(img) == d["image"]
    cl0 = orc.initialize_clusters(img, d["K"])
    labels, cl, lab, pre = orc.slic_iterate(img, cl0, stages=True)
    assert digest(lab) == d["lab"]
    assert digest(pre) == d["prelabels"]
    assert digest(orc.to_int16(labels)) == d["labels_int16"]
    assert digest(orc.cluster_table(cl)) == d["clusters_f32"]
    assert digest(cl) == d["clusters_raw"]


def test_heap_select_keeps_the_k_largest_and_matches_reference_on_ties(golden_small):
    # property: the kept set has the K largest areas (ties may be resolved either way);
    # exact tie behaviour is covered by the 4K digest (area tie at the cut) in the GPU/ref tests
    rng = np.random.default_rng(7)
    for _ in range(20):
        n = int(rng.integers(2, 200))
        K = int(rng.integers(1, n))
        area = rng.integers(1, 12, n).astype(np.int32)
        kept = orc.heap_select(np.arange(n), K, area)[:K]
        assert len(set(kept.tolist())) == K
        cut = np.sort(area)[::-1][K - 1]
        assert (area[kept] >= cut).all()
        assert (np.sort(area[kept])[::-1] == np.sort(area)[::-1][:K]).all()


def _topdown_sift(h, area, hole, ln, value):
    """The form of libstdc++'s __adjust_heap the device replays (fast_slic_amd/csrc/cca.hip, tie_adjust_heap): walk down the
    preferred children (smaller area, right child on a tie) and stop at the first child whose area exceeds the value's."""
    cur, va = hole, area[value]
    while True:
        if cur < (ln - 1) // 2:
            r = 2 * (cur + 1)
            child = r - 1 if area[h[r]] > area[h[r - 1]] else r
        elif (ln & 1) == 0 and cur == (ln - 2) // 2:
            child = 2 * cur + 1
        else:
            break
        if area[h[child]] > va:
            break
        h[cur] = h[child]
        cur = child
    h[cur] = value


def _topdown_heap_select(M, K, area):
    h = list(range(K))
    for p in range((K - 2) // 2, -1, -1) if K >= 2 else []:
        _topdown_sift(h, area, p, K, h[p])
    for i in range(K, M):
        if area[i] > area[h[0]]:
            _topdown_sift(h, area, 0, K, i)
    return sorted(h)


def test_topdown_sift_keeps_exactly_what_libstdcxx_heap_select_keeps():
    # The device resolves an area tie at the top-K cut with a top-down sift that stops early instead of libstdc++'s
    # walk-to-the-leaf-and-push-back; both must keep the same components on every input, ties included.
    rng = np.random.RandomState(1)
    for trial in range(1500):
        M = int(rng.randint(2, 400))
        K = int(rng.randint(1, M))
        style = trial % 4
        if style == 0:
            area = rng.randint(1, 6, M)
        elif style == 1:
            area = rng.randint(1, 50, M)
        elif style == 2:
            area = np.sort(rng.randint(1, 20, M))
        else:
            area = np.sort(rng.randint(1, 20, M))[::-1].copy()
        area = area.astype(np.int32)
        got = _topdown_heap_select(M, K, area)
        exp = sorted(orc.heap_select(np.arange(M, dtype=np.int32), K, area)[:K].tolist())
        assert got == exp, (trial, M, K)


def _sparse_heap_select(M, K, area):
    """The rule the device replays since round 5 (fast_slic_amd/csrc/cca.hip, tie_heap_select): with a* the K-th largest area, a
    sift -- of __make_heap and of __pop_heap alike -- stops at the first preferred child whose area exceeds min(value's area, a*),
    and the kept set is "area > a*, or still in the heap" (one bit per candidate).  The arrangement of the entries above a* then
    differs from libstdc++'s; the kept set must not."""
    astar = int(np.sort(area)[::-1][K - 1])
    h = list(range(K))
    inheap = [i < K for i in range(M)]

    def sift(hole, value):
        cur, vlim = hole, min(int(area[value]), astar)
        while True:
            if cur < (K - 1) // 2:
                r = 2 * (cur + 1)
                child = r - 1 if area[h[r]] > area[h[r - 1]] else r
            elif (K & 1) == 0 and cur == (K - 2) // 2:
                child = 2 * cur + 1
            else:
                break
            if area[h[child]] > vlim:
                break
            h[cur] = h[child]
            cur = child
        h[cur] = value

    for p in range((K - 2) // 2, -1, -1) if K >= 2 else []:
        sift(p, h[p])
    for i in range(K, M):
        if area[i] > area[h[0]]:
            assert area[h[0]] <= astar          # the root is never an entry above a*: fewer than K of those exist
            inheap[h[0]] = False
            inheap[i] = True
            sift(0, i)
    return [i for i in range(M) if area[i] > astar or inheap[i]]


def test_sparse_sift_keeps_exactly_what_libstdcxx_heap_select_keeps():
    rng = np.random.RandomState(2)
    for trial in range(3000):
        M = int(rng.randint(2, 400))
        K = int(rng.randint(1, M))
        style = trial % 6
        if style == 0:
            area = rng.randint(1, 6, M)
        elif style == 1:
            area = rng.randint(1, 50, M)
        elif style == 2:
            area = np.sort(rng.randint(1, 20, M))
        elif style == 3:
            area = np.sort(rng.randint(1, 20, M))[::-1].copy()
        elif style == 4:
            area = rng.randint(1, 1000, M)
        else:                                       # the bench frames' shape: M a little above K, most areas far above the cut
            K = max(1, M - int(rng.randint(1, 40)))
            area = np.where(rng.rand(M) < 0.15, rng.randint(1, 4, M), rng.randint(50, 60, M))
        area = area.astype(np.int32)
        got = _sparse_heap_select(M, K, area)
        exp = sorted(orc.heap_select(np.arange(M, dtype=np.int32), K, area)[:K].tolist())
        assert got == exp, (trial, M, K)


@pytest.mark.skipif(not ref.available(), reason="oracle/_ref not built (no /root/reference here)")
def test_oracle_equals_live_reference():
    rng = np.random.default_rng(3)
    for trial in range(6):
        H, W = int(rng.integers(40, 200)), int(rng.integers(40, 200))
        K = int(rng.integers(4, 60))
        var = "ABCD"[trial % 4]
        img = variant(var, H, W, seed=trial)
        kw = dict(compactness=float(rng.choice([5.0, 10.0, 40.0])),
                  min_size_factor=float(rng.choice([0.0, 0.25, 0.5])),
                  subsample_stride=int(rng.choice([1, 2, 3])), max_iter=int(rng.integers(0, 6)))
        cl0 = ref.initialize_clusters(img, K)
        r = ref.slic_iterate(img, cl0, stages=True, num_threads=2, arch="x64/avx2", **kw)
        o = orc.slic_iterate(img, cl0, stages=True, **kw)
        for nm, a, b in zip(("labels", "clusters", "lab", "prelabels"), o, r):
            if nm == "clusters":
                assert a.tobytes() == b.tobytes(), (trial, kw)
            else:
                assert np.array_equal(a, b), describe_mismatch("%d/%s %s" % (trial, nm, kw), a, b)
    # warm start: second call re-uses moved centres and re-samples colours (src/context.cpp:128-135)
    img1, img2 = variant("A", 120, 160, 0), variant("A", 120, 160, 1)
    cl0 = ref.initialize_clusters(img1, 30)
    _, c_r = ref.slic_iterate(img1, cl0)
    _, c_o = orc.slic_iterate(img1, cl0)
    l_r, c_r2 = ref.slic_iterate(img2, c_r)
    l_o, c_o2 = orc.slic_iterate(img2, c_o)
    assert np.array_equal(l_r, l_o) and c_r2.tobytes() == c_o2.tobytes()


@pytest.mark.skipif(not ref.available(), reason="oracle/_ref not built (no /root/reference here)")
def test_oracle_equals_live_reference_4k_tie_at_cut(golden_digests):
    d = golden_digests["C3_2160x3840_k6000"]      # area tie at the top-K cut
    img = variant("A", d["H"], d["W"])
    cl0 = orc.initialize_clusters(img, d["K"])
    labels, cl = orc.slic_iterate(img, cl0)
    assert digest(orc.to_int16(labels)) == d["labels_int16"]
    assert digest(cl) == d["clusters_raw"]
