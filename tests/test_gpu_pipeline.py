"""GPU parity at the benchmark's own operating point, the device-side top-K tie, and concurrent callers (-m gpu).

* the submit / drain pipeline exactly as bench.py drives it: 8 x 1280x720 K=1600 per submission (seeds 0..7), 6 slots, dynamic batching,
  hipGraph replay, several rounds, EVERY label map and EVERY Cluster byte against the oracle; then the 64 frames of
  BASELINE.json configs[3] (seeds 0..63) and one group of eight 3840x2160 frames
* area ties at the top-K cut (src/cca.cpp:225-228) are resolved on the device by replaying libstdc++'s
  __heap_select; label maps built to tie massively are checked against the oracle (whose heap restatement
  tests/test_oracle.py pins against the live std::partial_sort) and against the live reference when present
* four Python threads drive four models through the process-wide engine at once (the reference allows that:
  cfast_slic.pyx:188-193 releases the GIL, per-call Context) and every result is bit-equal to the oracle
"""
import os
import threading

import numpy as np
import pytest

from oracle import oracle as orc
from oracle import ref
from fast_slic_amd import _binding as B
from fast_slic_amd import Engine, Slic, make_params
from fast_slic_amd.synth import variant
from util import describe_mismatch, cluster_fields_equal

pytestmark = pytest.mark.gpu


def _expected(frames, K):
    out = []
    for f in frames:
        labels, cl = orc.slic_iterate(f, orc.initialize_clusters(f, K))
        out.append((labels, cl))
    return out


def _run_pipeline(H, W, K, seeds, group, nslots, rounds, expect_replay=True, min_size_factor=0.25, max_iter=10, host_topk=False, engine=None, batching=0):
    """Submit `rounds` times the groups cut from `seeds` through fslic_hip_pipeline_submit on `nslots` slots; every
    submission owns its label maps and cluster blocks, all of them are compared with the oracle after the drain."""
    import torch
    p = make_params(max_iter, 10.0, min_size_factor, 3)
    frames = [variant("A", H, W, seed=s) for s in seeds]
    exp = [orc.slic_iterate(f, orc.initialize_clusters(f, K), max_iter=max_iter, min_size_factor=min_size_factor) for f in frames]
    cl0 = np.zeros((len(frames), K), B.CLUSTER_DTYPE)
    lib = B.load_library()
    for i, f in enumerate(frames):
        lib.fslic_hip_initialize_clusters(H, W, K, f.ctypes.data, cl0[i].ctypes.data)
    e = engine or Engine(0, nslots)
    try:
        if batching: e.pipeline_batching(batching)
        d_rgb = [torch.from_numpy(f).cuda() for f in frames]
        torch.cuda.synchronize()
        subs = []
        for r in range(rounds):
            for g0 in range(0, len(frames), group):
                idx = list(range(g0, min(g0 + group, len(frames))))
                d_lab = [torch.full((H, W), -7, dtype=torch.int16, device="cuda") for _ in idx]
                cls = cl0[idx].copy()
                a_rgb = e.pointer_array([d_rgb[i].data_ptr() for i in idx])
                a_lab = e.pointer_array([t.data_ptr() for t in d_lab])
                a_cls = e.pointer_array([cls[j].ctypes.data for j in range(len(idx))])
                torch.cuda.synchronize()
                e.pipeline_submit(a_rgb, a_cls, a_lab, len(idx), H, W, K, p)
                subs.append((r, idx, d_lab, cls, (a_rgb, a_lab, a_cls)))
        tot = e.pipeline_drain()
        assert tot["frames"] == rounds * len(frames)
        if batching: assert tot["groups"] <= len(subs)
        else: assert tot["groups"] == len(subs)
        if host_topk: assert tot["host_topk_frames"] > 0, tot
        else: assert tot["host_topk_frames"] == 0, tot      # ties are resolved on the device
        assert tot["device_ms"] > 0.0
        for r, idx, d_lab, cls, _keep in subs:
            for j, i in enumerate(idx):
                got = d_lab[j].cpu().numpy().view(np.uint16)
                assert np.array_equal(got, exp[i][0]), describe_mismatch("round %d seed %d" % (r, seeds[i]), got, exp[i][0])
                assert cls[j].tobytes() == exp[i][1].tobytes(), "round %d seed %d: %s" % (r, seeds[i], "; ".join(cluster_fields_equal(cls[j], exp[i][1])))
        if expect_replay:
            modes = [e.last_launch_mode(s) for s in range(nslots)]
            assert 2 in modes or os.environ.get("FSLIC_GRAPH") == "0", modes      # slots that served more than two groups replay their graph (unless graphs are switched off)
    finally:
        if engine is None: e.close()


def test_bench_operating_point_8x720p_six_slots_batching():
    # exactly bench.py's timed loop: one submission = seeds 0..7 at 1280x720 K=1600; 6 slots (bench.py's default) with dynamic
    # batching of up to two submissions per launch group; 60 submissions so that slots reach graph replay for the 16-frame form;
    # seeds 2 and 4 carry an area tie at the top-K cut
    _run_pipeline(720, 1280, 1600, list(range(8)), 8, 6, 60, batching=16)


def test_pipeline_without_batching_8x720p_eight_slots():
    _run_pipeline(720, 1280, 1600, list(range(8)), 8, 8, 32)


def test_dynamic_batching_serves_several_submissions_as_one_group():
    # bench.py's mode: submissions of 8 frames, a slot thread gathers up to two of them into one group of 16 (every launch then
    # covers 16 frames); every submission's label maps and Cluster bytes against the oracle, fewer groups than submissions
    import torch
    e = Engine(0, 3)
    try:
        _run_pipeline(720, 1280, 1600, list(range(8)), 8, 3, 12, engine=e, expect_replay=False, batching=16)
        # other geometries and submission sizes through the same engine (its arenas are re-carved)
        _run_pipeline(360, 640, 400, list(range(6)), 2, 3, 8, engine=e, expect_replay=False, batching=16)
        _run_pipeline(720, 1280, 1600, list(range(8)), 8, 3, 6, engine=e, expect_replay=False, batching=16)
    finally:
        e.close()


def test_threshold_zero_takes_the_host_topk_path_while_other_slots_record_graphs():
    # min_size_factor = 0: every component is a candidate (src/cca.cpp:14-20 with thres 0), more than the device sorts, so the
    # frames go through the host's partial_sort: while the other slots direct-launch, record and replay their groups
    _run_pipeline(360, 640, 400, list(range(6)), 2, 3, 4, min_size_factor=0.0, host_topk=True)


def test_one_engine_serves_groups_of_different_iteration_counts():
    # the bin slots carry generation stamps, max_iter + 3 per group: a group of fewer iterations after one of more must not
    # find the earlier group's stamps inside its own range (it did: stale slots taken for live, a memory fault at 720p)
    e = Engine(0, 2)
    try:
        for mi in (10, 5, 0, 12, 1, 10):
            _run_pipeline(360, 640, 400, list(range(4)), 4, 2, 10, max_iter=mi, engine=e)      # replayed from the fourth group of a slot on
        _run_pipeline(720, 1280, 1600, list(range(4)), 4, 2, 3, max_iter=10, engine=e, expect_replay=False)
        _run_pipeline(720, 1280, 1600, list(range(4)), 4, 2, 3, max_iter=4, engine=e, expect_replay=False)
    finally:
        e.close()


def test_config4_all_64_frames():
    # BASELINE.json configs[3]: 64 frames (seeds 0..63) of 1280x720 K=1600, 8 per group
    _run_pipeline(720, 1280, 1600, list(range(64)), 8, 6, 1, expect_replay=False)      # (8 groups on 6 slots: which slot serves a third one is the scheduler's business)


def test_one_group_of_eight_4k_frames():
    _run_pipeline(2160, 3840, 1600, list(range(8)), 8, 2, 5)


# ---- the other operating points bench.py prints: same pipeline, same slots, same batching, every frame --
def test_bench_operating_point_c3_four_4k_k6000_frames_six_slots():
    # BASELINE.json configs[2] as bench.py's `c3_k6000` leg runs it: 4 x 3840x2160 K=6000 per submission, 6 slots with batching
    # (4 + 4 + .. frames gathered up to 16 per launch group), graph replay; seed 0 carries an area tie at the top-K cut
    # (a[K-1] = a[K] = 374), resolved on the device inside the pipeline; 225 seeds pile up on the image centre
    _run_pipeline(2160, 3840, 6000, list(range(4)), 4, 6, 36, batching=16)      # 36 submissions: 9+ groups of up to 16 frames, slots reach graph replay


def test_bench_operating_point_eight_4k_frames_six_slots_batching_off_and_on():
    # bench.py's `roofline_4k` leg: 8 x 3840x2160 K=1600 per submission on 6 slots; without batching and with (16 frames per group)
    _run_pipeline(2160, 3840, 1600, list(range(8)), 8, 6, 12)
    _run_pipeline(2160, 3840, 1600, list(range(8)), 8, 6, 12, batching=16)


def test_host_memory_calls_after_batching_reserve():
    # fslic_hip_pipeline_batching(16) raises the arena reserve; the host-memory entry points stage their frames into
    # the arena BEFORE group_begin, so the reserve must be applied when they carve (it was applied later: staged frames lost)
    H, W, K = 240, 320, 150
    imgs = [variant("A", H, W, seed=s) for s in range(5)]
    exp = [orc.slic_iterate(im, orc.initialize_clusters(im, K)) for im in imgs]
    e = Engine(0, 2)
    try:
        e.pipeline_batching(16)
        p = make_params(10, 10.0, 0.25, 3)
        cl = orc.initialize_clusters(imgs[0], K).view(B.CLUSTER_DTYPE).copy()
        labels = e.iterate(imgs[0], cl, p)                                        # fresh engine, first call on its slot
        assert np.array_equal(labels, exp[0][0]) and cl.tobytes() == exp[0][1].tobytes()
        cls = [orc.initialize_clusters(im, K).view(B.CLUSTER_DTYPE).copy() for im in imgs]
        outs = [np.zeros((H, W), np.uint16) for _ in imgs]
        e.iterate_batch([im.ctypes.data for im in imgs], cls, [o.ctypes.data for o in outs], H, W, p, False)
        for i in range(len(imgs)):
            assert np.array_equal(outs[i], exp[i][0]), "frame %d" % i
            assert cls[i].tobytes() == exp[i][1].tobytes(), "frame %d clusters" % i
    finally:
        e.close()


def test_synchronous_calls_wait_while_the_pipeline_occupies_every_slot():
    # a slot serving the submit / drain queue collects its group itself; a synchronous call from another thread must
    # wait for it (the header says so), not fail with "every slot owns an unfinished asynchronous group"
    import torch
    H, W, K = 360, 640, 400
    frames = [variant("A", H, W, seed=s) for s in range(4)]
    exp = [orc.slic_iterate(f, orc.initialize_clusters(f, K)) for f in frames]
    small = variant("B", 120, 160)
    exp_small = orc.slic_iterate(small, orc.initialize_clusters(small, 40))
    e = Engine(0, 2)
    errors = []
    try:
        p = make_params(10, 10.0, 0.25, 3)
        d_rgb = [torch.from_numpy(f).cuda() for f in frames]
        torch.cuda.synchronize()
        stop = threading.Event()

        def sync_caller():
            try:
                while not stop.is_set():
                    cl = orc.initialize_clusters(small, 40).view(B.CLUSTER_DTYPE).copy()
                    labels = e.iterate(small, cl, p)
                    if not np.array_equal(labels, exp_small[0]) or cl.tobytes() != exp_small[1].tobytes():
                        errors.append("synchronous call differs")
            except Exception as ex:                          # noqa: BLE001
                errors.append("synchronous caller: %r" % (ex,))

        th = threading.Thread(target=sync_caller)
        th.start()
        subs = []
        for r in range(40):
            d_lab = [torch.empty((H, W), dtype=torch.int16, device="cuda") for _ in frames]
            cls = np.stack([orc.initialize_clusters(f, K).view(B.CLUSTER_DTYPE) for f in frames]).copy()
            arrs = (e.pointer_array([t.data_ptr() for t in d_rgb]), e.pointer_array([cls[j].ctypes.data for j in range(4)]),
                    e.pointer_array([t.data_ptr() for t in d_lab]))
            e.pipeline_submit(arrs[0], arrs[1], arrs[2], 4, H, W, K, p)
            subs.append((d_lab, cls, arrs))
        e.pipeline_drain()
        stop.set()
        th.join()
        for d_lab, cls, _ in subs:
            for j in range(4):
                assert np.array_equal(d_lab[j].cpu().numpy().view(np.uint16), exp[j][0]) and cls[j].tobytes() == exp[j][1].tobytes()
    finally:
        e.close()
    assert not errors, "\n".join(errors[:5])


# ---- the area tie at the top-K cut, on the device ---------------------------------------------------------------
def _tie_maps():
    rng = np.random.RandomState(5)
    maps = []
    # (a) a chequerboard of 8x8 cells with random labels: hundreds of components of area 64 (and multiples where equal
    #     neighbours merge): almost every cut falls inside a tie
    for (H, W, cell, nlab) in [(256, 256, 8, 40), (200, 328, 4, 7), (300, 300, 10, 500)]:
        gy, gx = (H + cell - 1) // cell, (W + cell - 1) // cell
        lab = rng.randint(0, nlab, (gy, gx)).astype(np.uint16)
        maps.append(np.repeat(np.repeat(lab, cell, 0), cell, 1)[:H, :W].copy())
    # (b) horizontal stripes of equal length
    m = np.zeros((128, 512), np.uint16)
    m[:] = (np.arange(512) // 16)[None, :] + 32 * (np.arange(128) % 2)[:, None]
    maps.append(m)
    # (c) noise: thousands of tiny components, threshold 1 keeps all of them as candidates
    maps.append(rng.randint(0, 3, (96, 160)).astype(np.uint16))
    return maps


@pytest.mark.parametrize("mi", range(5))
def test_device_resolves_area_ties_like_partial_sort(engine, mi):
    m = _tie_maps()[mi]
    ncomp_hint = [1, 2, 5, 17, 64, 200, 333, 700, 1500, 4000]
    for thr in (0, 1, 16, 64):
        for K in ncomp_hint:
            exp = orc.enforce_connectivity(m, K, thr)
            got = engine.enforce_connectivity(m, K, thr)
            assert np.array_equal(got, exp), describe_mismatch("map %d K=%d thr=%d" % (mi, K, thr), got, exp)
            if ref.available() and hasattr(ref, "enforce_connectivity"):
                live = ref.enforce_connectivity(m, K, thr)
                assert np.array_equal(got, live), describe_mismatch("map %d K=%d thr=%d vs the live reference" % (mi, K, thr), got, live)


# ---- concurrent callers ------------------------------------------------------------------------------------------
def test_four_threads_four_models_concurrently():
    shapes = [(240, 320, 150), (300, 400, 200), (480, 640, 200), (200, 264, 120)]
    imgs = [[variant("AB"[r % 2], H, W, seed=10 * t + r) for r in range(6)] for t, (H, W, K) in enumerate(shapes)]
    exp = []
    for t, (H, W, K) in enumerate(shapes):
        cl = None
        per = []
        for img in imgs[t]:                                  # warm start chain, like a video stream (clusters carry over)
            cl = orc.initialize_clusters(img, K) if cl is None else cl
            labels, cl = orc.slic_iterate(img, cl)
            per.append((orc.to_int16(labels), cl.copy()))
        exp.append(per)
    errors = []
    start = threading.Barrier(len(shapes))

    def work(t):
        try:
            H, W, K = shapes[t]
            s = Slic(num_components=K, compactness=10)
            start.wait()
            for r, img in enumerate(imgs[t]):
                got = s.iterate(img)
                if not np.array_equal(got, exp[t][r][0]):
                    errors.append("thread %d frame %d: labels differ" % (t, r))
                if s.slic_model.cluster_array.tobytes() != exp[t][r][1].tobytes():
                    errors.append("thread %d frame %d: clusters differ" % (t, r))
                if '"iterate"' not in s.slic_model.last_timing_report:
                    errors.append("thread %d frame %d: no timing report" % (t, r))
        except Exception as ex:                              # noqa: BLE001 - reported below
            errors.append("thread %d: %r" % (t, ex))

    threads = [threading.Thread(target=work, args=(t,)) for t in range(len(shapes))]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    assert not errors, "\n".join(errors)


def test_more_threads_than_slots_queue_up():
    # 8 threads on an engine with 2 slots: calls beyond the slot count wait for a slot instead of racing
    H, W, K = 160, 200, 60
    imgs = [variant("A", H, W, seed=s) for s in range(8)]
    exp = [orc.slic_iterate(im, orc.initialize_clusters(im, K)) for im in imgs]
    e = Engine(0, 2)
    p = make_params(10, 10.0, 0.25, 3)
    errors = []

    def work(t):
        try:
            for _ in range(3):
                cl = orc.initialize_clusters(imgs[t], K).view(B.CLUSTER_DTYPE).copy()
                labels = e.iterate(imgs[t], cl, p)
                if not np.array_equal(labels, exp[t][0]) or cl.tobytes() != exp[t][1].tobytes():
                    errors.append("thread %d differs" % t)
        except Exception as ex:                              # noqa: BLE001
            errors.append("thread %d: %r" % (t, ex))

    try:
        threads = [threading.Thread(target=work, args=(t,)) for t in range(8)]
        for th in threads:
            th.start()
        for th in threads:
            th.join()
    finally:
        e.close()
    assert not errors, "\n".join(errors)


def test_stage_entry_points_refuse_a_slot_with_a_group_in_flight():
    import torch
    H, W, K = 120, 160, 40
    img = variant("A", H, W)
    e = Engine(0, 1)
    try:
        p = make_params(10, 10.0, 0.25, 3)
        d = torch.from_numpy(img).cuda()
        out = torch.empty((H, W), dtype=torch.int16, device="cuda")
        cl = orc.initialize_clusters(img, K).view(B.CLUSTER_DTYPE).copy()
        torch.cuda.synchronize()
        a_rgb, a_lab, a_cl = e.pointer_array([d.data_ptr()]), e.pointer_array([out.data_ptr()]), e.pointer_array([cl.ctypes.data])
        e.submit_group(0, a_rgb, a_cl, a_lab, 1, H, W, K, p)
        with pytest.raises(ValueError):                      # the only slot owns an uncollected group
            e.enforce_connectivity(np.zeros((8, 8), np.uint16), 4, 0)
        with pytest.raises(ValueError):
            e.rgb_to_lab(img)
        e.wait_group(0)
        exp = orc.slic_iterate(img, orc.initialize_clusters(img, K))
        assert np.array_equal(out.cpu().numpy().view(np.uint16), exp[0]) and cl.tobytes() == exp[1].tobytes()
        assert e.enforce_connectivity(np.zeros((8, 8), np.uint16), 4, 0).shape == (8, 8)
    finally:
        e.close()


def test_enforce_connectivity_rejects_more_labels_than_uint16():
    e = Engine(0, 1)
    try:
        with pytest.raises(ValueError):
            e.enforce_connectivity(np.zeros((8, 8), np.uint16), 65537, 0)
    finally:
        e.close()


def test_parity_with_the_cluster_pass_always_fused():
    """By default the cluster pass is fused into the assign kernel only where a launch does not fill the chip (group.cpp); the
    larger operating points therefore never see it.  FSLIC_FUSEBIN=2 (read when the library is loaded: a separate interpreter)
    fuses it everywhere: BASELINE sizes incl. the 225-seed pile-up of C3 (clusters hidden behind an identical predecessor re-enter
    the bins one per pass), the 16-frame pipeline and a group of 4K frames must be bit-equal all the same."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, FSLIC_FUSEBIN="2")
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-m", "gpu", os.path.join(root, "tests", "test_gpu_parity.py"),
                        os.path.join(root, "tests", "test_gpu_pipeline.py"), "-k",
                        "baseline_sizes or small_golden or option or bench_operating_point_8x720p or one_group_of_eight_4k or warm or c3_four"],
                       cwd=root, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=1500)
    tail = r.stdout.decode()[-2500:]
    assert r.returncode == 0 and " passed" in tail, tail


@pytest.mark.parametrize("switch", ["FSLIC_FUSEBIN=0", "FSLIC_GROUP=1", "FSLIC_GROUP=16", "FSLIC_GRAPH=0", "FSLIC_POISON=1", "FSLIC_HOST_TIMING=1"])
def test_parity_under_every_switch_the_library_reads(switch):
    """Every environment switch of the shipping library (grep getenv fast_slic_amd/csrc: FSLIC_FUSEBIN, FSLIC_GROUP, FSLIC_GRAPH,
    FSLIC_POISON, FSLIC_HOST_TIMING; read once when the library is loaded, hence a separate interpreter) against the oracle: the
    BASELINE sizes (C2, C3, C3b), the batch entry points FSLIC_GROUP cuts into groups (64 frames of C4, device-resident and host
    batches), a replayed pipeline, and the overflow / redo paths on a poisoned arena."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    name, value = switch.split("=")
    env = dict(os.environ, **{name: value})
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-m", "gpu", os.path.join(root, "tests", "test_gpu_parity.py"),
                        os.path.join(root, "tests", "test_gpu_pipeline.py"), "-k",
                        "baseline_sizes or config4 or device_resident_and_batch or poisoned or pipeline_of_replayed"],
                       cwd=root, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=1500)
    tail = r.stdout.decode()[-2500:]
    assert r.returncode == 0 and " passed" in tail, tail



@pytest.mark.parametrize("stride,n", [(1, 8), (2, 12), (3, 16)])
def test_sixteen_row_fused_pass_on_the_2d_table_at_every_stride(stride, n):
    """Launches of more than 2048 eight-row blocks with the 2-D spatial table (S <= 40) take the fused passes with 16 rows per wavefront
    (launch_assign, assign.hip; its own table `tabs16`): 8 / 12 / 16 frames of 1280x720 K=1600 in ONE group reach that at subsample_stride
    1 / 2 / 3 (3680 / 2880 / 2560 blocks), the bench only at 3.  Every frame against the oracle, twice (the second group replays)."""
    import torch
    H, W, K = 720, 1280, 1600
    frames = [variant("A", H, W, seed=40 + i) for i in range(n)]
    p = make_params(10, 10.0, 0.25, stride)
    exp = [orc.slic_iterate(f, orc.initialize_clusters(f, K), subsample_stride=stride) for f in frames]
    e = Engine(0, 1)
    try:
        d_rgb = [torch.from_numpy(f).cuda() for f in frames]
        d_lab = [torch.empty((H, W), dtype=torch.int16, device="cuda") for _ in frames]
        torch.cuda.synchronize()
        a_rgb = e.pointer_array([t.data_ptr() for t in d_rgb])
        a_lab = e.pointer_array([t.data_ptr() for t in d_lab])
        for rep in range(3):
            cls = [orc.initialize_clusters(f, K) for f in frames]
            a_cls = e.pointer_array([c.ctypes.data for c in cls])
            e.submit_group(0, a_rgb, a_cls, a_lab, n, H, W, K, p)
            e.wait_group(0)
            assert e.last_group_frames(0) == n
            for i in range(n):
                got = d_lab[i].cpu().numpy().view(np.uint16)
                assert np.array_equal(got, exp[i][0]), describe_mismatch("stride %d rep %d frame %d" % (stride, rep, i), got, exp[i][0])
                assert cls[i].tobytes() == exp[i][1].tobytes(), (stride, rep, i)
    finally:
        e.close()
