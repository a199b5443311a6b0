"""lib/nms counterpart.  CPU: the oracle restatement and the product's numpy / host-C++ paths against
what the REFERENCE's lib/nms/nms.py produced (tests/golden/nms.npz, oracle/make_golden.py::nms_case).  GPU: the HIP
suppression-mask + sweep kernels (buctd_nms) against the same golden keep lists - bit-exact index lists."""
import os

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "nms.npz")


def _db(g, tag):
    k = g[f"kpts_{tag}"]
    return [{"keypoints": k[i].reshape(-1, 3), "score": float(g[f"scores_{tag}"][i]), "area": float(g[f"areas_{tag}"][i])}
            for i in range(k.shape[0])]


def test_oracle_and_host_paths_match_reference_golden():
    from oracle import nms as on
    import buctd_amd.nms.nms as pn
    g = np.load(GOLD)
    for tag in "abc":
        d, thr, keep = g[f"boxes_{tag}"], float(g[f"thr_{tag}"]), g[f"keep_{tag}"].tolist()
        assert on.nms(d, thr) == keep and on.nms(d, thr, strict=False) == keep
        assert [int(i) for i in pn.nms(d, thr)] == keep
        assert pn.cpu_nms(d, thr) == keep                      # host C++ (buctd_cpu_nms)
        assert [int(i) for i in pn.py_nms_wrapper(thr)(d)] == keep and pn.cpu_nms_wrapper(thr)(d) == keep
    assert pn.nms(np.zeros((0, 5), np.float32), 0.5) == [] and pn.cpu_nms(np.zeros((0, 5), np.float32), 0.5) == []
    for tag in "pq":
        sig = g[f"sigmas_{tag}"] if f"sigmas_{tag}" in g.files else None
        kp, ar = g[f"kpts_{tag}"], g[f"areas_{tag}"]
        for impl in (on, pn):
            assert np.allclose(impl.oks_iou(kp[0], kp[1:], ar[0], ar[1:], sig), g[f"oks_{tag}"], atol=1e-14)
            assert np.allclose(impl.oks_iou(kp[0], kp[1:], ar[0], ar[1:], sig, 0.3), g[f"oksvis_{tag}"], atol=1e-14)
            db = _db(g, tag)
            assert [int(i) for i in impl.oks_nms(db, 0.6, sig)] == g[f"oksnms_{tag}"].tolist()
            assert [int(i) for i in impl.oks_nms(db, 0.6, sig, 0.3)] == g[f"oksnmsvis_{tag}"].tolist()
            assert [int(i) for i in impl.soft_oks_nms(db, 0.6, sig)] == g[f"softnms_{tag}"].tolist()
    assert pn.oks_nms([], 0.5) == [] and len(pn.soft_oks_nms([], 0.5)) == 0
    # oks_merge: a far-away pose is appended, a duplicate is not
    db = _db(g, "p")
    far = {"keypoints": db[0]["keypoints"] + np.array([5000.0, 5000.0, 0.0]), "score": 1.0, "area": 3000.0}
    merged = pn.oks_merge([dict(db[0]), far], list(db[1:4]) + [dict(db[0])])
    assert len(merged) == 5 and merged[-1] is far


@pytest.mark.gpu
def test_hip_nms_matches_reference_golden(dev):
    import buctd_amd.nms.nms as pn
    from oracle import nms as on
    g = np.load(GOLD)
    for tag in "abc":
        d, thr = g[f"boxes_{tag}"], float(g[f"thr_{tag}"])
        assert pn.gpu_nms(d, thr, 0) == g[f"keep_{tag}"].tolist(), f"HIP nms differs from the reference on set {tag}"
        assert pn.gpu_nms_wrapper(thr, 0)(d) == g[f"keep_{tag}"].tolist()
    for n, seed, thr in ((1, 7, 0.5), (64, 8, 0.3), (129, 9, 0.7), (3000, 10, 0.4)):   # ragged / multi-block sizes
        d = on.make_boxes(n, seed, size=200.0 if n < 3000 else 1500.0)
        assert pn.gpu_nms(d, thr, 0) == on.nms(d, thr), f"HIP nms vs oracle, n = {n}"
    assert pn.gpu_nms(np.zeros((0, 5), np.float32), 0.5, 0) == []
