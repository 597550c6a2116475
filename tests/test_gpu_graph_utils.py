"""Superpixel-graph utilities on the GPU (graph.hip; src/fast-slic.cpp): get_connectivity, get_mask_density,
cluster_density_to_mask through the C ABI.  Integer work, so bit-exact: against the committed fixtures
(tests/golden/graph_cases.npz, generated from the unmodified reference) and against the reference run live (oracle/_ref)
on label maps of this engine's own Slic.iterate, on adversarial maps (noise: every cluster touches every other and the
12-neighbour cut-off decides; labels >= K), with host and with device-resident planes."""
import os
import numpy as np
import pytest

from fast_slic_amd import make_params, SlicModel
from fast_slic_amd import _binding as B
from fast_slic_amd.synth import variant
from oracle import ref

pytestmark = pytest.mark.gpu
GOLD = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "graph_cases.npz"))
CASES = sorted({k.split("/")[0] for k in GOLD.files})
needs_ref = pytest.mark.skipif(not ref.available(), reason="oracle/_ref (the reference build) is not present")


def clusters_of(case):
    return np.ascontiguousarray(GOLD[case + "/clusters"]).view(B.CLUSTER_DTYPE).reshape(-1)


@pytest.mark.parametrize("case", CASES)
def test_golden_vectors(engine, case):
    labels, mask, cl = GOLD[case + "/labels"], GOLD[case + "/mask"], clusters_of(case)
    H, W = labels.shape
    K = cl.shape[0]
    if case + "/conn_num" in GOLD.files:
        num, nb = engine.get_connectivity(labels, H, W, K)
        np.testing.assert_array_equal(num, GOLD[case + "/conn_num"])
        np.testing.assert_array_equal(nb, GOLD[case + "/conn_nb"])
    dens = engine.get_mask_density(cl, labels, mask, H, W)
    np.testing.assert_array_equal(dens, GOLD[case + "/density"])
    np.testing.assert_array_equal(engine.cluster_density_to_mask(labels, dens, H, W), GOLD[case + "/broadcast"])


def check_against_reference(engine, labels, cl, mask, what, connectivity=True):
    H, W = labels.shape
    K = cl.shape[0]
    if connectivity:
        num, nb = engine.get_connectivity(labels, H, W, K)
        rnum, rnb = ref.get_connectivity(labels, K)
        np.testing.assert_array_equal(num, rnum, err_msg=what)
        np.testing.assert_array_equal(nb, rnb, err_msg=what)
    dens = engine.get_mask_density(cl, labels, mask, H, W)
    np.testing.assert_array_equal(dens, ref.get_mask_density(cl, labels, mask), err_msg=what)
    np.testing.assert_array_equal(engine.cluster_density_to_mask(labels, dens, H, W), ref.cluster_density_to_mask(cl, labels, dens), err_msg=what)


@needs_ref
def test_label_maps_of_the_engine_against_the_live_reference(engine):
    rng = np.random.default_rng(99)
    for trial, (H, W, K) in enumerate([(97, 131, 35), (240, 320, 200), (480, 640, 600), (720, 1280, 1600), (301, 1, 3), (1, 64, 4), (2, 2, 2)]):
        img = variant("ABCD"[trial % 4], H, W, seed=trial)
        cl = np.zeros(K, B.CLUSTER_DTYPE)
        B.load_library().fslic_hip_initialize_clusters(H, W, K, img.ctypes.data, cl.ctypes.data)
        labels = engine.iterate(np.ascontiguousarray(img), cl, make_params(5, 10.0, 0.25, 3))
        mask = (rng.integers(0, 256, (H, W)) * (rng.random((H, W)) < 0.6)).astype(np.uint8)
        check_against_reference(engine, labels, cl, mask, "%dx%d K=%d" % (H, W, K), connectivity=bool((labels < K).all()))


@needs_ref
def test_adversarial_maps_against_the_live_reference(engine):
    rng = np.random.default_rng(5)
    for trial in range(24):
        H, W = int(rng.integers(2, 150)), int(rng.integers(2, 190))
        K = int(rng.integers(1, 400))
        kind = trial % 4
        if kind == 0:      # noise: far more distinct pairs than 12 K; exercises the cut-off and the growth of the pair table
            labels = rng.integers(0, K, (H, W)).astype(np.uint16)
        elif kind == 1:    # coarse blocks with noise sprinkled in
            labels = ((np.arange(H)[:, None] // 7) * 13 + np.arange(W)[None, :] // 9).astype(np.uint16) % K
            m = rng.random((H, W)) < 0.1
            labels[m] = rng.integers(0, K, int(m.sum()))
        elif kind == 2:    # one cluster adjacent to many (a frame around blocks): its list fills up first
            labels = ((np.arange(H)[:, None] // 4) * 50 + np.arange(W)[None, :] // 4).astype(np.uint16) % K
            labels[::4] = 0
            labels[:, ::4] = 0
        else:              # few labels, long shared borders
            labels = (rng.integers(0, min(K, 3), (H // 8 + 1, W // 8 + 1)).repeat(8, 0).repeat(8, 1)[:H, :W]).astype(np.uint16)
        cl = np.zeros(K, B.CLUSTER_DTYPE)
        cl["number"] = np.arange(K)
        cl["num_members"] = rng.integers(0, 50, K)
        mask = rng.integers(0, 256, (H, W)).astype(np.uint8)
        check_against_reference(engine, np.ascontiguousarray(labels), cl, mask, "trial %d kind %d %dx%d K=%d" % (trial, kind, H, W, K))
        # labels >= K (unassigned pixels): the density utilities skip them (the reference's connectivity reads out of bounds there)
        holes = labels.copy()
        holes[rng.random((H, W)) < 0.1] = 0xFFFF
        check_against_reference(engine, holes, cl, mask, "holes, trial %d" % trial, connectivity=False)


def test_device_resident_planes_give_the_same_answer(engine):
    torch = pytest.importorskip("torch")
    case = "slic_240x320_k300"
    labels, mask, cl = GOLD[case + "/labels"], GOLD[case + "/mask"], clusters_of(case)
    H, W = labels.shape
    K = cl.shape[0]
    d_lab = torch.from_numpy(labels.view(np.int16).copy()).cuda()
    d_mask = torch.from_numpy(mask.copy()).cuda()
    d_out = torch.zeros((H, W), dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    num, nb = engine.get_connectivity(d_lab.data_ptr(), H, W, K)
    np.testing.assert_array_equal(num, GOLD[case + "/conn_num"])
    np.testing.assert_array_equal(nb, GOLD[case + "/conn_nb"])
    dens = engine.get_mask_density(cl, d_lab.data_ptr(), d_mask.data_ptr(), H, W)
    np.testing.assert_array_equal(dens, GOLD[case + "/density"])
    engine.cluster_density_to_mask(d_lab.data_ptr(), dens, H, W, out=d_out.data_ptr())
    np.testing.assert_array_equal(d_out.cpu().numpy(), GOLD[case + "/broadcast"])


def test_model_surface_mirrors_the_reference(engine):
    # cfast_slic.pyx:262-324: SlicModel.get_connectivity / get_knn_connectivity / get_mask_density / broadcast_density_to_mask
    case = "slic_96x128_k40"
    labels, mask, cl = GOLD[case + "/labels"], GOLD[case + "/mask"], clusters_of(case)
    m = SlicModel(cl.shape[0])
    m.cluster_array[:] = cl
    a = labels.view(np.int16)
    conn = m.get_connectivity(a).tolist()
    assert conn == [[int(v) for v in GOLD[case + "/conn_nb"][k, :GOLD[case + "/conn_num"][k]]] for k in range(cl.shape[0])]
    dens = m.get_mask_density(mask, a)
    np.testing.assert_array_equal(dens, GOLD[case + "/density"])
    np.testing.assert_array_equal(m.broadcast_density_to_mask(dens, a), GOLD[case + "/broadcast"])
    with pytest.raises(ValueError):
        m.get_mask_density(mask[:-1], a)
    with pytest.raises(ValueError):
        m.broadcast_density_to_mask(dens[:-1], a)


@needs_ref
def test_connectivity_of_a_large_noise_map_grows_its_pair_table_quickly(engine):
    # 700x900 noise over 20 000 labels: ~1.9 million distinct adjacent pairs against a first table of a million entries.  Until round 6 every
    # thread that found the table full probed ALL of it before giving up and nobody else stopped (a 2072x2348 map did not finish in 13
    # minutes: tests/fuzz_parity.py, graph mode, FUZZ_SHAPES=big); now the first thread to find it half full ends the launch.
    import time
    rng = np.random.default_rng(8)
    H, W, K = 700, 900, 20000
    labels = np.ascontiguousarray(rng.integers(0, K, (H, W)).astype(np.uint16))
    t0 = time.time()
    num, nb = engine.get_connectivity(labels, H, W, K)
    took = time.time() - t0
    rnum, rnb = ref.get_connectivity(labels, K)
    np.testing.assert_array_equal(num, rnum)
    np.testing.assert_array_equal(nb, rnb)
    assert took < 30.0, "get_connectivity took %.1f s" % took
