"""CPU tests of the drop-in boundary: the C-ABI library loads and exports every symbol that
include/fslic_hip.h declares; host-only entry points behave like the reference; argument errors map
to the reference's exception types.  No kernel is launched here."""
import os
import re

import numpy as np
import pytest

import fast_slic_amd
from fast_slic_amd import _binding as B
from oracle import oracle as orc
from fast_slic_amd.synth import variant

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    src = open(os.path.join(ROOT, "include", "fslic_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(fslic_hip_\w+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    lib = B.load_library()
    names = declared_symbols()
    assert len(names) >= 14
    for n in names:
        assert hasattr(lib, n), n
    assert sorted(names) == sorted(B.EXPORTS)


def test_cluster_abi_is_32_bytes_with_reference_offsets():
    d = B.CLUSTER_DTYPE
    assert d.itemsize == 32
    offs = {n: d.fields[n][1] for n in d.names}
    assert offs == dict(y=0, x=4, r=8, g=12, b=16, a=20, number=24, is_active=26, is_updatable=27, num_members=28)


def test_params_struct_layout_and_abi_word():
    """include/fslic_hip.h: 64 bytes, `abi` where a testing flag (0.2) and later `variant` used to live, `variant` behind it; the
    ctypes mirror fills the ABI word the header defines (a caller built against another layout is refused by validate())."""
    import ctypes as C
    P = B.Params
    assert C.sizeof(P) == 64
    offs = {n: getattr(P, n).offset for n, _ in P._fields_}
    assert offs["max_iter"] == 0 and offs["debug_mode"] == 36 and offs["abi"] == 40 and offs["variant"] == 44 and offs["reserved"] == 48
    hdr = open(os.path.join(ROOT, "include", "fslic_hip.h")).read()
    assert int(re.search(r"#define FSLIC_PARAMS_ABI (0x[0-9a-fA-F]+)", hdr).group(1), 16) == B.PARAMS_ABI
    assert B.make_params(10, 10.0, 0.25, 3).abi == B.PARAMS_ABI
    assert b"0.3" in B.load_library().fslic_hip_version()


@pytest.mark.parametrize("H,W,K", [(480, 640, 200), (2160, 3840, 6000), (37, 53, 7), (10, 10, 100), (5, 300, 17)])
def test_initialize_clusters_matches_oracle(H, W, K):
    img = variant("A", H, W)
    cl = np.zeros(K, B.CLUSTER_DTYPE)
    rc = B.load_library().fslic_hip_initialize_clusters(H, W, K, img.ctypes.data, cl.ctypes.data)
    assert rc == 0
    assert cl.tobytes() == orc.initialize_clusters(img, K).tobytes()


def test_supported_archs_surface():
    assert fast_slic_amd.get_supported_archs() == ["hip/gfx950"]
    assert fast_slic_amd.supported_archs == ("hip/gfx950",)
    assert fast_slic_amd.is_supported_arch("hip/gfx950")
    assert not fast_slic_amd.is_supported_arch("x64/avx2")


def test_slic_model_argument_errors_match_reference():
    with pytest.raises(NotImplementedError):        # cfast_slic.pyx:21-22
        B.SlicModel(10, "x64/avx2")
    with pytest.raises(ValueError):                 # cfast_slic.pyx:24-25
        B.SlicModel(65534)
    with pytest.raises(ValueError):                 # cfast_slic.pyx:26-27
        B.SlicModel(0)


def test_no_cpu_fallback_without_gpu():
    if B.device_count() > 0:
        pytest.skip("a GPU is visible")
    with pytest.raises(RuntimeError):
        fast_slic_amd.Slic(num_components=10)
    with pytest.raises(RuntimeError):
        fast_slic_amd.Engine(0, 1)


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "fast_slic_amd")
    for dirpath, _, files in os.walk(pkg):
        for fn in files:
            if fn.endswith((".py", ".cpp", ".hip", ".h")):
                text = open(os.path.join(dirpath, fn)).read()
                assert "oracle" not in text.replace("no CPU fallback", ""), os.path.join(dirpath, fn)


def test_the_library_reads_no_untested_switch():
    """grep getenv over the library's sources == the switches tests/test_gpu_pipeline.py::test_parity_under_every_switch_the_library_reads (and FSLIC_FUSEBIN=2 next to it) runs."""
    import glob
    root = ROOT
    found = set()
    for fn in glob.glob(os.path.join(root, "fast_slic_amd", "csrc", "*")):
        if os.path.isfile(fn):
            found |= set(re.findall(r'getenv\("(\w+)"\)', open(fn).read()))
    assert found == {"FSLIC_FUSEBIN", "FSLIC_GROUP", "FSLIC_GRAPH", "FSLIC_POISON", "FSLIC_HOST_TIMING"}, found
