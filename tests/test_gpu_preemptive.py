"""GPU tests of the preemptive mode (src/preemptive.h; `Slic(preemptive=True)`) through the C ABI: bit-equal to the
unmodified reference (fixtures in tests/golden/preemptive_cases.npz from tests/golden/make_golden_preemptive.py) --
pre-connectivity labels, final labels and every Cluster field including the is_updatable counters."""
import json
import os

import numpy as np
import pytest

from fast_slic_amd import Slic, make_params
from fast_slic_amd import _binding as B
from fast_slic_amd.synth import variant
from util import describe_mismatch, cluster_fields_equal

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_preemptive_bit_equal_to_reference_fixtures(engine):
    cases = np.load(os.path.join(ROOT, "tests", "golden", "preemptive_cases.npz"), allow_pickle=False)
    names = sorted({k.split("/")[0] for k in cases.files if k.endswith("/shape") and not k.startswith("rd_")})
    assert len(names) >= 9
    for name in names:
        H, W, K = (int(v) for v in cases[name + "/shape"])
        kw = json.loads(str(cases[name + "/kwargs"]))
        img = np.ascontiguousarray(variant(str(cases[name + "/variant"]), H, W))
        cl = np.zeros(K, B.CLUSTER_DTYPE)
        B._check(B.load_library().fslic_hip_initialize_clusters(H, W, K, img.ctypes.data, cl.ctypes.data))
        p = make_params(kw.get("max_iter", 10), kw.get("compactness", 10.0), 0.25, kw.get("subsample_stride", 3), True, True,
                        preemptive=True, preemptive_thres=float(cases[name + "/thres"]))
        labels = engine.iterate(img, cl, p)
        pre = engine.last_prelabels(H, W)
        assert np.array_equal(pre, cases[name + "/prelabels"]), describe_mismatch(name + "/prelabels", pre, cases[name + "/prelabels"])
        msgs = cluster_fields_equal(cl, cases[name + "/clusters"])
        assert not msgs, name + ": " + "; ".join(msgs)
        assert np.array_equal(labels, cases[name + "/labels"]), describe_mismatch(name + "/labels", labels, cases[name + "/labels"])


KIND = {"standard": B.VARIANT_REALDIST, "l2": B.VARIANT_REALDIST_L2, "noq": B.VARIANT_REALDIST_NOQ}


def test_preemptive_realdist_variants_bit_equal_to_reference_fixtures(engine):
    # BaseContext<float>::iterate is the same template (src/context.cpp:152-181): SlicRealDist / L2 / NoQ with preemptive=True
    cases = np.load(os.path.join(ROOT, "tests", "golden", "preemptive_cases.npz"), allow_pickle=False)
    names = sorted({k.split("/")[0] for k in cases.files if k.endswith("/shape") and k.startswith("rd_")})
    assert len(names) >= 13
    for name in names:
        H, W, K = (int(v) for v in cases[name + "/shape"])
        kw = json.loads(str(cases[name + "/kwargs"]))
        img = np.ascontiguousarray(variant(str(cases[name + "/variant"]), H, W))
        cl = np.zeros(K, B.CLUSTER_DTYPE)
        B._check(B.load_library().fslic_hip_initialize_clusters(H, W, K, img.ctypes.data, cl.ctypes.data))
        p = make_params(kw.get("max_iter", 10), kw.get("compactness", 10.0), 0.25, kw.get("subsample_stride", 3), True, kw.get("manhattan", True),
                        preemptive=True, preemptive_thres=float(cases[name + "/thres"]), variant=KIND[str(cases[name + "/kind"])])
        labels = engine.iterate(img, cl, p)
        pre = engine.last_prelabels(H, W)
        assert np.array_equal(pre, cases[name + "/prelabels"]), describe_mismatch(name + "/prelabels", pre, cases[name + "/prelabels"])
        msgs = cluster_fields_equal(cl, cases[name + "/clusters"])
        assert not msgs, name + ": " + "; ".join(msgs)
        assert np.array_equal(labels, cases[name + "/labels"]), describe_mismatch(name + "/labels", labels, cases[name + "/labels"])


def test_preemptive_facade_and_unsupported_variants(engine):
    img = variant("A", 120, 160)
    out = Slic(num_components=40, compactness=10, preemptive=True, preemptive_thres=0.05).iterate(img)
    assert out.dtype == np.int16 and out.min() >= 0
    cl = np.zeros(40, B.CLUSTER_DTYPE)
    B._check(B.load_library().fslic_hip_initialize_clusters(120, 160, 40, img.ctypes.data, cl.ctypes.data))
    with pytest.raises(NotImplementedError):
        engine.iterate(np.ascontiguousarray(img), cl, make_params(10, 10.0, 0.25, 3, preemptive=True, variant=B.VARIANT_LSC))
