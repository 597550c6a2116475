"""Helpers shared by the parity tests."""
import hashlib
import json

import numpy as np


def digest(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()[:16]


def small_case_names(npz):
    return sorted({k.split("/")[0] for k in npz.files if k.endswith("/image")})


def cca_case_names(npz):
    return sorted({k.split("/")[0] for k in npz.files if k.endswith("/in")})


def case_kwargs(npz, name):
    return json.loads(str(npz[name + "/kwargs"]))


def describe_mismatch(name, got, exp, limit=8):
    got = np.asarray(got)
    exp = np.asarray(exp)
    if got.shape != exp.shape:
        return "%s: shape %s != %s" % (name, got.shape, exp.shape)
    bad = np.argwhere(got != exp)
    if bad.size == 0:
        return ""
    lines = ["%s: %d of %d entries differ" % (name, len(bad), got.size)]
    for idx in bad[:limit]:
        t = tuple(int(i) for i in idx)
        lines.append("   at %s got %s expected %s" % (t, got[t], exp[t]))
    return "\n".join(lines)


def cluster_fields_equal(a, b):
    """Every field of Cluster except the never-written `a` (src/fast-slic-common.h:16)."""
    msgs = []
    for f in ("y", "x", "r", "g", "b", "number", "is_active", "is_updatable", "num_members"):
        if not np.array_equal(a[f], b[f]):
            bad = np.nonzero(a[f] != b[f])[0]
            msgs.append("cluster field %s differs at %d entries, first k=%d got %s expected %s"
                        % (f, len(bad), bad[0], a[f][bad[0]], b[f][bad[0]]))
    return msgs


# ---- partition agreement (LSC: f32 path, tolerance-based parity) ----------------
def best_overlap(a, b):
    """Fraction of pixels whose segment in `a` lies in the segment of `b` that it overlaps most."""
    a = np.asarray(a).astype(np.int64).ravel()
    b = np.asarray(b).astype(np.int64).ravel()
    nb = int(b.max()) + 1
    u, c = np.unique(a * nb + b, return_counts=True)
    ua = u // nb
    best = np.zeros(int(ua.max()) + 1, np.int64)
    np.maximum.at(best, ua, c)
    return float(best.sum()) / a.size


def boundary_mask(a):
    a = np.asarray(a)
    m = np.zeros(a.shape, bool)
    m[:, :-1] |= a[:, :-1] != a[:, 1:]
    m[:-1, :] |= a[:-1, :] != a[1:, :]
    return m


def boundary_iou(a, b):
    ma, mb = boundary_mask(a), boundary_mask(b)
    return float((ma & mb).sum()) / max(1, int((ma | mb).sum()))
