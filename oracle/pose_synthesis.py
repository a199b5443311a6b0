"""TEST INFRASTRUCTURE - generative pose synthesis (reference lib/dataset/pose_synthesis.py: synthesize_pose 779-817,
synthesize_pose_coco 505-775, synthesize_pose_crowdpose 234-501), SURVEY 8f row f2.

The reference draws from numpy's / Python's global generators, which cannot be matched; what CAN be matched is the
distribution.  This restatement keeps the reference's sampling scheme - per joint five error types (jitter, miss,
inversion, swap, good), each proposing candidates on a ring around a source key point and keeping those that are far
enough from the other sources, one survivor drawn uniformly, then the type drawn from the renormalised table - but
drives it with a counter-based generator (uniform(seed, person, joint, stream, index)), the same one the HIP kernel uses:
  * "one survivor uniformly out of the survivors of N iid candidates" = count the survivors, draw a rank, take the
    survivor of that rank (identical distribution, incl. the probability of no survivor);
  * the miss type keeps all survivors around the first source and a with-replacement sub-sample of size n//4 of the
    others before drawing uniformly from the union = source s with probability proportional to n_0 resp. n_s//4, then a
    uniform survivor of that source.
Quirks kept: class tables indexed by joint id (crowdpose joints 12/13 inherit the jitter class of joint 11 through the
reference's un-reset variable); the index 1 + n_swap is treated as 'the inversion source' even when no inversion source
exists; crowdpose output visibility is 0, coco 1.
Pinned two ways by oracle/make_golden.py: (a) distributionally against the imported reference function (class
frequencies per joint on a fixed scene), (b) the HIP kernel must reproduce this module sample by sample."""
import math

import numpy as np

M64 = (1 << 64) - 1
N_CAND = 500

COCO = dict(
    sigmas=np.array([.26, .25, .25, .35, .35, .79, .79, .72, .72, .62, .62, 1.07, 1.07, .87, .87, .89, .89]) / 10.0,
    symmetry=[(1, 2), (3, 4), (5, 6), (7, 8), (9, 10), (11, 12), (13, 14), (15, 16)],
    jitter_cls=[0, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 2, 2, 0, 0, 0, 0],
    miss_cls=[0, 0, 0, 0, 0, 1, 1, 2, 2, 2, 2, 2, 2, 2, 2, 1, 1],
    inv_cls=[0, 0, 0, 0, 0, 1, 1, 1, 1, 1, 1, 2, 2, 2, 2, 2, 2],
    swap_cls=[0, 0, 0, 0, 0, 1, 1, 1, 1, 1, 1, 2, 2, 2, 2, 2, 2], out_vis=1.0)
CROWDPOSE = dict(
    sigmas=np.array([.79, .79, .72, .72, .62, .62, 1.07, 1.07, .87, .87, .89, .89, .79, .79]) / 10.0,
    symmetry=[(0, 1), (2, 3), (4, 5), (6, 7), (8, 9), (10, 11)],
    jitter_cls=[1, 1, 1, 1, 1, 1, 2, 2, 0, 0, 0, 0, 0, 0],      # 12, 13: whatever joint 11 left in jitter_prob
    miss_cls=[1, 1, 2, 2, 2, 2, 2, 2, 1, 1, 2, 2, 0, 0],
    inv_cls=[1, 1, 1, 1, 1, 1, 2, 2, 2, 2, 2, 2, 0, 0],
    swap_cls=[1, 1, 1, 1, 1, 1, 2, 2, 2, 2, 2, 2, 0, 0], out_vis=0.0)
JITTER_P = [[0.15, 0.20, 0.25], [0.10, 0.15, 0.20]]                 # [nv <= 10 | else][class]
MISS_P = [[0.15, 0.20, 0.25], [0.10, 0.13, 0.15], [0.02, 0.05, 0.10]]  # [nv <= 5 | <= 10 | else][class]
INV_P = [0.01, 0.03, 0.06]
SWAP_P = [[0.02, 0.15, 0.10], [0.01, 0.06, 0.03]]                   # [crowded | else][class]


def tables(dataset):
    return COCO if dataset == "coco" else CROWDPOSE


def uniform(seed, person, joint, stream, index):
    """Counter-based uniforms in [0, 1) with 53 random bits: splitmix64 of a packed key.  index may be an array."""
    key = (((int(person) * 64 + int(joint)) * 64 + int(stream)) << 24)
    z = (np.asarray(index, dtype=np.uint64) + np.uint64(key & M64)) + np.uint64(1)
    with np.errstate(over="ignore"):
        z = np.uint64(seed & M64) + z * np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return (z >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)


def _ring(seed, b, j, stream, n, cx, cy, r_lo, r_hi):
    idx = np.arange(n)
    ang = uniform(seed, b, j, stream, 2 * idx) * (2 * math.pi)
    r = r_lo + (r_hi - r_lo) * uniform(seed, b, j, stream, 2 * idx + 1)
    return cx + r * np.cos(ang), cy + r * np.sin(ang), r


def synthesize_pose(dataset, joints, estimated, near, area, num_overlap, seed, person=0):
    """joints, estimated [K, 3]; near [M, K, 3]; returns [K, 3]."""
    T = tables(dataset)
    K = joints.shape[0]
    var = (T["sigmas"] * 2) ** 2
    d10, d50, d85 = (np.sqrt(-2 * area * var * np.log(ks)) for ks in (0.10, 0.50, 0.85))
    synth = np.array(joints, dtype=np.float64).copy()
    for j in range(K):
        if joints[j, 2] == 0:
            synth[j] = estimated[j]
    nv = int(np.sum(joints[:, 2] > 0))
    # Known deviation from the reference (shared with the HIP kernel, synth.hip): every joint is synthesized from the
    # UNPERTURBED pose `synth`; the reference's in-place update makes the second joint of a symmetric pair see the first
    # one's perturbed position as its inversion source (pose_synthesis.py:271).  Parity with the reference is therefore
    # distributional (make_golden.py:pose_synthesis_case), sample-exact only between this twin and the kernel.
    pair_of = {}
    for q, w in T["symmetry"]:
        pair_of[q], pair_of[w] = w, q
    out = synth.copy()
    for j in range(K):
        pair = pair_of.get(j)
        src = [synth[j, :2]]
        swap = [near[m, j, :2] for m in range(near.shape[0]) if near[m, j, 2] > 0]
        src += swap
        has_inv = pair is not None and joints[pair, 2] > 0
        if has_inv:
            src.append(synth[pair, :2])
        swapinv = [near[m, pair, :2] for m in range(near.shape[0]) if near[m, pair, 2] > 0] if pair is not None else []
        src += swapinv
        src = np.array(src, dtype=np.float64)
        ns = len(src)
        skip = 1 + len(swap)                     # 'the inversion source', whether or not one exists

        def survivors(stream, s, n, r_lo, r_hi, others, thr):
            x, y, r = _ring(seed, person, j, stream, n, src[s, 0], src[s, 1], r_lo, r_hi)
            ok = np.ones(n, dtype=bool)
            for i in others:
                dist = np.sqrt((src[i, 0] - x) ** 2 + (src[i, 1] - y) ** 2)
                ok &= dist > (r if thr is None else thr)
            return x, y, ok

        def pick(stream, lists):
            """lists: [(x, y, ok, weight)]: source with probability ~ weight, then a uniform survivor."""
            total = sum(wt for *_, wt in lists)
            if total == 0:
                return np.zeros(3)
            t = int(uniform(seed, person, j, stream, 0) * total)
            for x, y, ok, wt in lists:
                if t < wt:
                    n = int(ok.sum())
                    k = int(uniform(seed, person, j, stream, 1) * n)
                    sel = np.nonzero(ok)[0][k]
                    return np.array([x[sel], y[sel], 1.0])
                t -= wt
            raise AssertionError

        # jitter (stream 0), miss (streams 1.., pick 40), inversion (41/42), swap (43.., pick 60), good (61/62)
        x, y, ok = survivors(0, 0, N_CAND, d85[j], d50[j], [i for i in range(ns) if i != 0], None)
        s_jit = pick(30, [(x, y, ok, int(ok.sum()))])
        lists = []
        for s in range(ns):
            x, y, ok = survivors(1 + s, s, 4 * N_CAND, d50[j], d10[j], [i for i in range(ns) if i != s], d50[j])
            n = int(ok.sum())
            lists.append((x, y, ok, n if s == 0 else n // 4))
        s_miss = pick(40, lists)
        s_inv = np.zeros(3)
        if has_inv:
            x, y, ok = survivors(41, skip, N_CAND, 0.0, d50[j], [i for i in range(ns) if i != skip], None)
            s_inv = pick(42, [(x, y, ok, int(ok.sum()))])
        s_swap = np.zeros(3)
        if len(swap) > 0 or len(swapinv) > 0:
            lists = []
            guards = [i for i in (0, skip) if i < ns]
            for s in range(ns):
                if s == 0 or s == skip:
                    continue
                x, y, ok = survivors(43 + s, s, N_CAND, 0.0, d50[j], guards, None)
                lists.append((x, y, ok, int(ok.sum())))
            s_swap = pick(60, lists)
        x, y, ok = survivors(61, 0, N_CAND // 4, 0.0, d85[j], [i for i in range(ns) if i != 0], None)
        s_good = pick(62, [(x, y, ok, int(ok.sum()))])

        p_jit = JITTER_P[0 if nv <= 10 else 1][T["jitter_cls"][j]]
        p_miss = MISS_P[0 if nv <= 5 else (1 if nv <= 10 else 2)][T["miss_cls"][j]]
        p_inv = INV_P[T["inv_cls"][j]]
        crowded = (nv <= 10 and num_overlap > 0) or (nv <= 15 and num_overlap >= 3)
        p_swap = SWAP_P[0 if crowded else 1][T["swap_cls"][j]]
        p_good = 1 - (p_jit + p_miss + p_inv + p_swap)
        cands = [s_jit, s_miss, s_inv, s_swap, s_good]
        probs = [p if c[2] != 0 else 0.0 for p, c in zip([p_jit, p_miss, p_inv, p_swap, p_good], cands)]
        norm = probs[0] + probs[1] + probs[2] + probs[3] + probs[4]
        if norm == 0:
            out[j] = 0
            continue
        u = uniform(seed, person, j, 63, 0) * norm
        acc, chosen = 0.0, 4
        for t in range(5):
            acc += probs[t]
            if u < acc:
                chosen = t
                break
        while cands[chosen][2] == 0:            # u == norm to rounding: fall back to the last proposed type
            chosen -= 1
        out[j, :2] = cands[chosen][:2]
        out[j, 2] = T["out_vis"]
    return out


def classify(point, joints, estimated, near, area, dataset, j):
    """Which error type a synthesized point looks like (geometry only): 0 good, 1 jitter, 2 inversion, 3 swap, 4 miss.
    Used to compare distributions with the reference's output, which does not label its choice."""
    T = tables(dataset)
    var = (T["sigmas"] * 2) ** 2
    d50, d85 = (np.sqrt(-2 * area * var[j] * np.log(ks)) for ks in (0.50, 0.85))
    gt = joints[j, :2] if joints[j, 2] != 0 else estimated[j, :2]
    dist = float(np.hypot(*(point[:2] - gt)))
    if dist <= d85:
        return 0
    if dist <= d50:
        return 1
    pair = dict([(q, w) for q, w in T["symmetry"]] + [(w, q) for q, w in T["symmetry"]]).get(j)
    if pair is not None and joints[pair, 2] > 0 and np.hypot(*(point[:2] - joints[pair, :2])) <= d50:
        return 2
    for m in range(near.shape[0]):
        for jj in ([j] if pair is None else [j, pair]):
            if near[m, jj, 2] > 0 and np.hypot(*(point[:2] - near[m, jj, :2])) <= d50:
                return 3
    return 4


def make_scene(dataset, seed, n_near=2):
    """A fixed person with neighbours for the distribution checks."""
    K = 17 if dataset == "coco" else 14
    rng = np.random.RandomState(seed)
    joints = np.concatenate([rng.rand(K, 2) * np.array([120, 220]) + 60, np.ones((K, 1))], 1)
    joints[3, 2] = 0
    joints[3, :2] = 0
    est = joints.copy()
    est[:, :2] = joints[:, :2] + rng.randn(K, 2) * 3
    est[3, :2] = [100.0, 120.0]
    near = np.concatenate([rng.rand(n_near, K, 2) * np.array([160, 240]) + 40, np.ones((n_near, K, 1))], 2)
    near[0, 5, 2] = 0
    return joints, est, near, 160.0 * 240.0
