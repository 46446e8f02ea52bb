"""ctypes binding of the CPU oracle (oracle/libvslam_oracle.so).  TEST INFRASTRUCTURE ONLY.

Importers allowed: tests/, __graft_entry__.smoke(), bench.py's cpu_baseline leg, tools/make_golden.py.
The product package (visual-slam_amd/) never imports this module.
"""
import ctypes as C
import os
import subprocess
from pathlib import Path

import numpy as np

_DIR = Path(__file__).resolve().parent
_LIB = None

u8p = C.POINTER(C.c_uint8)
i32p = C.POINTER(C.c_int32)
u32p = C.POINTER(C.c_uint32)
u64p = C.POINTER(C.c_uint64)
i64p = C.POINTER(C.c_int64)
f32p = C.POINTER(C.c_float)
f64p = C.POINTER(C.c_double)


def build(force=False):
    so = _DIR / "libvslam_oracle.so"
    srcs = [_DIR / n for n in ("orc_keypoints.cpp", "orc_bow.cpp", "orc_ba.cpp", "orc_vo.cpp", "orc_orb.cpp", "orc_pgo.cpp", "vslam_oracle.h",
                               "rbrief_pattern.inc", "Makefile")]
    if force or not so.exists() or any(s.stat().st_mtime > so.stat().st_mtime for s in srcs):
        subprocess.run(["make", "-C", str(_DIR)], check=True, capture_output=True)
    return so


class BaProblem(C.Structure):
    _fields_ = [("n_cams", C.c_int32), ("n_lms", C.c_int32), ("n_obs", C.c_int32),
                ("cam_model", C.c_int32 * 2), ("poses", f64p), ("cam_fixed", u8p),
                ("cam_intr", i32p), ("intr", f64p), ("points", f64p), ("obs_cam", i32p),
                ("obs_lm", i32p), ("obs_uv", f64p)]


class BaOptions(C.Structure):
    _fields_ = [("use_huber", C.c_int32), ("huber_parameter", C.c_double),
                ("max_num_iterations", C.c_int32), ("verbosity", C.c_int32),
                ("num_threads", C.c_int32)]


class BaSummary(C.Structure):
    _fields_ = [("initial_cost", C.c_double), ("final_cost", C.c_double),
                ("iterations", C.c_int32), ("successful_steps", C.c_int32),
                ("termination", C.c_int32), ("linearize_ms", C.c_double),
                ("schur_ms", C.c_double), ("solve_ms", C.c_double), ("total_ms", C.c_double)]


def lib():
    global _LIB
    if _LIB is None:
        so = build()
        L = C.CDLL(str(so))
        L.orc_voc_load_text.restype = C.c_void_p
        L.orc_voc_load_text.argtypes = [C.c_char_p]
        L.orc_voc_free.argtypes = [C.c_void_p]
        L.orc_voc_info.argtypes = [C.c_void_p, i32p, i32p, i32p, i32p]
        L.orc_bow_score_l1.restype = C.c_double
        _LIB = L
    return _LIB


def _img(img):
    img = np.ascontiguousarray(img, dtype=np.uint8)
    assert img.ndim == 2
    return img, img.ctypes.data_as(u8p), img.shape[1], img.shape[0], C.c_size_t(img.strides[0])


def min_eig_response(img):
    img, p, w, h, pitch = _img(img)
    out = np.empty((h, w), np.float32)
    lib().orc_min_eig_response(p, w, h, pitch, out.ctypes.data_as(f32p))
    return out


def good_features(img, max_corners, quality=0.01, min_dist=8.0):
    img, p, w, h, pitch = _img(img)
    cap = max_corners if max_corners > 0 else (w * h) // 4 + 1
    xy = np.zeros((cap, 2), np.int32)
    n = lib().orc_good_features(p, w, h, pitch, int(max_corners), C.c_double(quality),
                                C.c_double(min_dist), xy.ctypes.data_as(i32p), None)
    return xy[:n].copy()


def detect_keypoints(img, num_features):
    img, p, w, h, pitch = _img(img)
    xy = np.zeros((max(num_features, 1), 2), np.float64)
    n = lib().orc_detect_keypoints(p, w, h, pitch, int(num_features), xy.ctypes.data_as(f64p))
    return xy[:n].copy()


def compute_angles(img, corners, rotate=True):
    img, p, w, h, pitch = _img(img)
    corners = np.ascontiguousarray(corners, np.float64)
    n = len(corners)
    ang = np.zeros(n, np.float64)
    lib().orc_compute_angles(p, w, h, pitch, corners.ctypes.data_as(f64p), n, int(rotate),
                             ang.ctypes.data_as(f64p))
    return ang


def patch_moments(img, corners):
    img, p, w, h, pitch = _img(img)
    corners = np.ascontiguousarray(corners, np.float64)
    n = len(corners)
    m01 = np.zeros(n, np.int64)
    m10 = np.zeros(n, np.int64)
    lib().orc_patch_moments(p, w, h, pitch, corners.ctypes.data_as(f64p), n,
                            m01.ctypes.data_as(i64p), m10.ctypes.data_as(i64p))
    return m01, m10


def compute_descriptors(img, corners, angles):
    img, p, w, h, pitch = _img(img)
    corners = np.ascontiguousarray(corners, np.float64)
    angles = np.ascontiguousarray(angles, np.float64)
    n = len(corners)
    desc = np.zeros((n, 4), np.uint64)
    lib().orc_compute_descriptors(p, w, h, pitch, corners.ctypes.data_as(f64p),
                                  angles.ctypes.data_as(f64p), n, desc.ctypes.data_as(u64p))
    return desc


def detect_describe(img, num_features, rotate=True):
    img, p, w, h, pitch = _img(img)
    cap = max(num_features, 1)
    xy = np.zeros((cap, 2), np.float64)
    ang = np.zeros(cap, np.float64)
    desc = np.zeros((cap, 4), np.uint64)
    n = lib().orc_detect_describe(p, w, h, pitch, int(num_features), int(rotate),
                                  xy.ctypes.data_as(f64p), ang.ctypes.data_as(f64p),
                                  desc.ctypes.data_as(u64p))
    return xy[:n].copy(), ang[:n].copy(), desc[:n].copy()


def match_descriptors(d1, d2, threshold=70, dist_2_best=1.2):
    d1 = np.ascontiguousarray(d1, np.uint64).reshape(-1, 4)
    d2 = np.ascontiguousarray(d2, np.uint64).reshape(-1, 4)
    pairs = np.zeros((max(len(d1), 1), 2), np.int32)
    n = lib().orc_match_descriptors(d1.ctypes.data_as(u64p), len(d1), d2.ctypes.data_as(u64p),
                                    len(d2), int(threshold), C.c_double(dist_2_best),
                                    pairs.ctypes.data_as(i32p))
    return pairs[:n].copy()


def bitset_to_bytes(desc):
    desc = np.ascontiguousarray(desc, np.uint64).reshape(-1, 4)
    out = np.zeros((len(desc), 32), np.uint8)
    for i in range(len(desc)):
        lib().orc_bitset_to_bytes(desc[i].ctypes.data_as(u64p), out[i].ctypes.data_as(u8p))
    return out


def bytes_to_bitset(b):
    b = np.ascontiguousarray(b, np.uint8).reshape(-1, 32)
    out = np.zeros((len(b), 4), np.uint64)
    for i in range(len(b)):
        lib().orc_bytes_to_bitset(b[i].ctypes.data_as(u8p), out[i].ctypes.data_as(u64p))
    return out


# ---- ORB front end of compute_bow_vector ([upstream] cv::ORB restated, parity unpinned)
def orb_level_sizes(w, h, nlevels=8):
    lw, lh = np.zeros(nlevels, np.int32), np.zeros(nlevels, np.int32)
    sc = np.zeros(nlevels, np.float32)
    lib().orc_orb_level_sizes(int(w), int(h), int(nlevels), lw.ctypes.data_as(i32p), lh.ctypes.data_as(i32p), sc.ctypes.data_as(f32p))
    return lw, lh, sc


def orb_level_quota(nfeatures, nlevels=8):
    q = np.zeros(nlevels, np.int32)
    lib().orc_orb_level_quota(int(nfeatures), int(nlevels), q.ctypes.data_as(i32p))
    return q


def orb_resize(src, dw, dh):
    src = np.ascontiguousarray(src, np.uint8)
    dst = np.zeros((dh, dw), np.uint8)
    lib().orc_orb_resize(src.ctypes.data_as(u8p), src.shape[1], src.shape[0], dst.ctypes.data_as(u8p), int(dw), int(dh))
    return dst


def orb_fast_score(img, x, y, thr=20):
    img = np.ascontiguousarray(img, np.uint8)
    return lib().orc_orb_fast_score(img.ctypes.data_as(u8p), img.shape[1], img.shape[0], int(x), int(y), int(thr))


def orb_gauss7(img):
    img = np.ascontiguousarray(img, np.uint8)
    out = np.zeros_like(img)
    lib().orc_orb_gauss7(img.ctypes.data_as(u8p), img.shape[1], img.shape[0], out.ctypes.data_as(u8p))
    return out


def orb_fast_atan2(y, x):
    f = lib().orc_orb_fast_atan2
    f.restype = C.c_float
    return float(f(C.c_float(y), C.c_float(x)))


def orb_detect_describe(img, nfeatures=1500):
    img = np.ascontiguousarray(img, np.uint8)
    cap = 2 * nfeatures + 64   # retainBest keeps ties beyond the quota
    kp = np.zeros((cap, 5), np.float32)
    desc = np.zeros((cap, 32), np.uint8)
    n = lib().orc_orb_detect_describe(img.ctypes.data_as(u8p), img.shape[1], img.shape[0], C.c_size_t(img.strides[0]),
                                      int(nfeatures), kp.ctypes.data_as(f32p), desc.ctypes.data_as(u8p), cap)
    return kp[:n].copy(), desc[:n].copy()


def project_landmarks(pose7, model, intr8, width, height, points, cam_z_threshold=0.1):
    pose7 = np.ascontiguousarray(pose7, np.float64)
    intr8 = np.ascontiguousarray(intr8, np.float64)
    points = np.ascontiguousarray(points, np.float64).reshape(-1, 3)
    n = len(points)
    uv = np.zeros((max(n, 1), 2), np.float64)
    idx = np.zeros(max(n, 1), np.int32)
    m = lib().orc_project_landmarks(pose7.ctypes.data_as(f64p), int(model), intr8.ctypes.data_as(f64p), int(width),
                                    int(height), points.ctypes.data_as(f64p), n, C.c_double(cam_z_threshold),
                                    uv.ctypes.data_as(f64p), idx.ctypes.data_as(i32p))
    return uv[:m].copy(), idx[:m].copy()


def find_matches_landmarks(kp_xy, kp_desc, proj_uv, proj_lm, lm_obs_start, obs_desc, max_dist_2d=20.0, threshold=70,
                           dist_2_best=1.2):
    kp_xy = np.ascontiguousarray(kp_xy, np.float64).reshape(-1, 2)
    kp_desc = np.ascontiguousarray(kp_desc, np.uint64).reshape(-1, 4)
    proj_uv = np.ascontiguousarray(proj_uv, np.float64).reshape(-1, 2)
    proj_lm = np.ascontiguousarray(proj_lm, np.int32)
    lm_obs_start = np.ascontiguousarray(lm_obs_start, np.int32)
    obs_desc = np.ascontiguousarray(obs_desc, np.uint64).reshape(-1, 4)
    pairs = np.zeros((max(len(kp_xy), 1), 2), np.int32)
    n = lib().orc_find_matches_landmarks(kp_xy.ctypes.data_as(f64p), kp_desc.ctypes.data_as(u64p), len(kp_xy),
                                         proj_uv.ctypes.data_as(f64p), proj_lm.ctypes.data_as(i32p), len(proj_uv),
                                         lm_obs_start.ctypes.data_as(i32p), obs_desc.ctypes.data_as(u64p),
                                         C.c_double(max_dist_2d), int(threshold), C.c_double(dist_2_best),
                                         pairs.ctypes.data_as(i32p))
    return pairs[:n].copy()


class Vocabulary:
    def __init__(self, path):
        self._h = lib().orc_voc_load_text(os.fsencode(str(path)))
        if not self._h:
            raise IOError("orc_voc_load_text failed: %s" % path)

    def info(self):
        v = [C.c_int32() for _ in range(4)]
        lib().orc_voc_info(self._h, *[C.byref(x) for x in v])
        return tuple(x.value for x in v)

    def transform(self, desc32, levelsup=4):
        desc32 = np.ascontiguousarray(desc32, np.uint8).reshape(-1, 32)
        n = len(desc32)
        ids = np.zeros(max(n, 1), np.uint32)
        vals = np.zeros(max(n, 1), np.float64)
        fn = np.zeros(max(n, 1), np.uint32)
        ff = np.zeros(max(n, 1), np.uint32)
        nnz, fvn = C.c_int32(), C.c_int32()
        lib().orc_bow_transform(C.c_void_p(self._h), desc32.ctypes.data_as(u8p), n, int(levelsup),
                                ids.ctypes.data_as(u32p), vals.ctypes.data_as(f64p), C.byref(nnz),
                                fn.ctypes.data_as(u32p), ff.ctypes.data_as(u32p), C.byref(fvn))
        return (ids[:nnz.value].copy(), vals[:nnz.value].copy(), fn[:fvn.value].copy(),
                ff[:fvn.value].copy())

    def __del__(self):
        if getattr(self, "_h", None):
            lib().orc_voc_free(C.c_void_p(self._h))
            self._h = None


def bow_score_l1(ids1, v1, ids2, v2):
    ids1 = np.ascontiguousarray(ids1, np.uint32)
    ids2 = np.ascontiguousarray(ids2, np.uint32)
    v1 = np.ascontiguousarray(v1, np.float64)
    v2 = np.ascontiguousarray(v2, np.float64)
    return lib().orc_bow_score_l1(ids1.ctypes.data_as(u32p), v1.ctypes.data_as(f64p), len(ids1),
                                  ids2.ctypes.data_as(u32p), v2.ctypes.data_as(f64p), len(ids2))


def _stream_args(ids, vals, ops):
    ids = np.ascontiguousarray(ids, np.uint32)
    vals = np.ascontiguousarray(vals, np.float64)
    ops = np.ascontiguousarray(ops, np.uint8)
    assert len(ids) == len(vals) == len(ops)
    return ids, vals, ops


def _bowvec_stream(fn, ids, vals, ops, norm):
    ids, vals, ops = _stream_args(ids, vals, ops)
    n = len(ids)
    oi = np.zeros(max(n, 1), np.uint32)
    ov = np.zeros(max(n, 1), np.float64)
    fn.restype = C.c_int
    m = fn(ids.ctypes.data_as(u32p), vals.ctypes.data_as(f64p), ops.ctypes.data_as(u8p), n, int(norm),
           oi.ctypes.data_as(u32p), ov.ctypes.data_as(f64p))
    return oi[:m].copy(), ov[:m].copy()


def _featvec_stream(fn, nodes, feats):
    nodes = np.ascontiguousarray(nodes, np.uint32)
    feats = np.ascontiguousarray(feats, np.uint32)
    n = len(nodes)
    on = np.zeros(max(n, 1), np.uint32)
    of = np.zeros(max(n, 1), np.uint32)
    fn.restype = C.c_int
    m = fn(nodes.ctypes.data_as(u32p), feats.ctypes.data_as(u32p), n, on.ctypes.data_as(u32p), of.ctypes.data_as(u32p))
    return on[:m].copy(), of[:m].copy()


def bowvec_stream(ids, vals, ops, norm):
    """The restatement of BowVector::addWeight / addIfNotExist / normalize replayed on an operation stream."""
    return _bowvec_stream(lib().orc_bowvec_stream, ids, vals, ops, norm)


def featvec_stream(nodes, feats):
    return _featvec_stream(lib().orc_featvec_stream, nodes, feats)


_REF = None


def ref_lib():
    """oracle/_ref/libdbow2_ref.so: the REFERENCE's own BowVector.cpp / FeatureVector.cpp (compiled where they lie by
    `make -C oracle ref`) behind oracle/ref_dbow2_driver.cpp.  Returns None when it is neither prebuilt nor buildable
    (no /root/reference on this machine)."""
    global _REF
    if _REF is None:
        so = _DIR / "_ref" / "libdbow2_ref.so"
        if Path("/root/reference/thirdparty/DBoW2_ORBSLAM/DBoW2/BowVector.cpp").exists():
            subprocess.run(["make", "-C", str(_DIR), "ref"], check=True, capture_output=True)
        if not so.exists():
            return None
        _REF = C.CDLL(str(so))
    return _REF


def ref_bowvec_stream(ids, vals, ops, norm):
    return _bowvec_stream(ref_lib().ref_bowvec_stream, ids, vals, ops, norm)


def ref_featvec_stream(nodes, feats):
    return _featvec_stream(ref_lib().ref_featvec_stream, nodes, feats)


def project(model, intr8, p3):
    intr8 = np.ascontiguousarray(intr8, np.float64)
    p3 = np.ascontiguousarray(p3, np.float64)
    uv = np.zeros(2)
    lib().orc_project(int(model), intr8.ctypes.data_as(f64p), p3.ctypes.data_as(f64p),
                      uv.ctypes.data_as(f64p))
    return uv


def ba_residual(model, pose7, point3, intr8, uv2):
    a = [np.ascontiguousarray(x, np.float64) for x in (pose7, point3, intr8, uv2)]
    r = np.zeros(2)
    lib().orc_ba_residual(int(model), a[0].ctypes.data_as(f64p), a[1].ctypes.data_as(f64p),
                          a[2].ctypes.data_as(f64p), a[3].ctypes.data_as(f64p),
                          r.ctypes.data_as(f64p))
    return r


def ba_residual_jacobian(model, pose7, point3, intr8, uv2):
    a = [np.ascontiguousarray(x, np.float64) for x in (pose7, point3, intr8, uv2)]
    r, Jp, Jl = np.zeros(2), np.zeros((2, 6)), np.zeros((2, 3))
    lib().orc_ba_residual_jacobian(int(model), a[0].ctypes.data_as(f64p), a[1].ctypes.data_as(f64p),
                                   a[2].ctypes.data_as(f64p), a[3].ctypes.data_as(f64p),
                                   r.ctypes.data_as(f64p), Jp.ctypes.data_as(f64p),
                                   Jl.ctypes.data_as(f64p))
    return r, Jp, Jl


def se3_plus(pose7, delta6):
    a = np.ascontiguousarray(pose7, np.float64)
    d = np.ascontiguousarray(delta6, np.float64)
    out = np.zeros(7)
    lib().orc_se3_plus(a.ctypes.data_as(f64p), d.ctypes.data_as(f64p), out.ctypes.data_as(f64p))
    return out


class BaArrays:
    """Owns contiguous numpy arrays of a flattened BA problem (see include/vslam_hip.h)."""

    def __init__(self, poses, cam_fixed, cam_intr, intr, points, obs_cam, obs_lm, obs_uv,
                 cam_model=(0, 0)):
        self.poses = np.ascontiguousarray(poses, np.float64).reshape(-1, 7).copy()
        self.cam_fixed = np.ascontiguousarray(cam_fixed, np.uint8).copy()
        self.cam_intr = np.ascontiguousarray(cam_intr, np.int32).copy()
        self.intr = np.ascontiguousarray(intr, np.float64).reshape(2, 8).copy()
        self.points = np.ascontiguousarray(points, np.float64).reshape(-1, 3).copy()
        self.obs_cam = np.ascontiguousarray(obs_cam, np.int32).copy()
        self.obs_lm = np.ascontiguousarray(obs_lm, np.int32).copy()
        self.obs_uv = np.ascontiguousarray(obs_uv, np.float64).reshape(-1, 2).copy()
        self.cam_model = tuple(int(m) for m in cam_model)

    def copy(self):
        return BaArrays(self.poses, self.cam_fixed, self.cam_intr, self.intr, self.points,
                        self.obs_cam, self.obs_lm, self.obs_uv, self.cam_model)

    def fill(self, st):
        st.n_cams, st.n_lms, st.n_obs = len(self.poses), len(self.points), len(self.obs_cam)
        st.cam_model[0], st.cam_model[1] = self.cam_model
        st.poses = self.poses.ctypes.data_as(f64p)
        st.cam_fixed = self.cam_fixed.ctypes.data_as(u8p)
        st.cam_intr = self.cam_intr.ctypes.data_as(i32p)
        st.intr = self.intr.ctypes.data_as(f64p)
        st.points = self.points.ctypes.data_as(f64p)
        st.obs_cam = self.obs_cam.ctypes.data_as(i32p)
        st.obs_lm = self.obs_lm.ctypes.data_as(i32p)
        st.obs_uv = self.obs_uv.ctypes.data_as(f64p)
        return st

    @property
    def n_free(self):
        return int((self.cam_fixed == 0).sum())


def _opts(use_huber=True, huber=1.0, max_iters=20, verbosity=0, threads=1):
    o = BaOptions()
    o.use_huber, o.huber_parameter, o.max_num_iterations = int(use_huber), float(huber), int(max_iters)
    o.verbosity, o.num_threads = int(verbosity), int(threads)
    return o


def ba_linearize(arr, use_huber=True, huber=1.0, lm_first=0, lm_count=-1):
    st = arr.fill(BaProblem())
    o = _opts(use_huber, huber)
    n = 6 * arr.n_free
    S = np.zeros((n, n))
    g = np.zeros(n)
    cost, nf = C.c_double(), C.c_int32()
    lib().orc_ba_linearize(C.byref(st), C.byref(o), int(lm_first), int(lm_count),
                           S.ctypes.data_as(f64p), g.ctypes.data_as(f64p), C.byref(cost), C.byref(nf))
    return S, g, cost.value


def bundle_adjust(arr, use_huber=True, huber=1.0, max_iters=20, verbosity=0, threads=1):
    """Optimises arr.poses / arr.points in place; returns the summary struct."""
    st = arr.fill(BaProblem())
    o = _opts(use_huber, huber, max_iters, verbosity, threads)
    s = BaSummary()
    lib().orc_bundle_adjust(C.byref(st), C.byref(o), C.byref(s))
    return s


def bundle_adjust_intrinsics(arr, use_huber=True, huber=1.0, max_iters=20, verbosity=0):
    """optimize_intrinsics = true: optimises arr.poses / arr.points / arr.intr in place; returns the summary struct."""
    st = arr.fill(BaProblem())
    o = _opts(use_huber, huber, max_iters, verbosity, 1)
    s = BaSummary()
    lib().orc_bundle_adjust_intrinsics(C.byref(st), C.byref(o), arr.intr.ctypes.data_as(f64p), C.byref(s))
    return s


def ba_residual_jacobian_intr(model, pose7, point3, intr8, uv2):
    a = [np.ascontiguousarray(x, np.float64) for x in (pose7, point3, intr8, uv2)]
    Ji = np.zeros((2, 8))
    lib().orc_ba_residual_jacobian_intr(int(model), a[0].ctypes.data_as(f64p), a[1].ctypes.data_as(f64p),
                                        a[2].ctypes.data_as(f64p), a[3].ctypes.data_as(f64p), Ji.ctypes.data_as(f64p))
    return Ji


# ---- pose graph optimisation ([upstream] Ceres + Sophus restated, parity unpinned)
class PgoProblem(C.Structure):
    _fields_ = [("n_nodes", C.c_int32), ("n_edges", C.c_int32), ("poses", f64p), ("node_fixed", u8p),
                ("edge_a", i32p), ("edge_b", i32p), ("edge_meas", f64p)]


class PgoArrays:
    """Flattened pose graph: poses [N, 7] (in/out), node_fixed [N] u8, edge_a / edge_b [E] i32, edge_meas [E, 6]."""

    def __init__(self, poses, node_fixed, edge_a, edge_b, edge_meas):
        self.poses = np.ascontiguousarray(poses, np.float64).reshape(-1, 7).copy()
        self.node_fixed = np.ascontiguousarray(node_fixed, np.uint8)
        self.edge_a = np.ascontiguousarray(edge_a, np.int32)
        self.edge_b = np.ascontiguousarray(edge_b, np.int32)
        self.edge_meas = np.ascontiguousarray(edge_meas, np.float64).reshape(-1, 6)

    def fill(self, st):
        st.n_nodes, st.n_edges = len(self.poses), len(self.edge_a)
        st.poses = self.poses.ctypes.data_as(f64p)
        st.node_fixed = self.node_fixed.ctypes.data_as(u8p)
        st.edge_a = self.edge_a.ctypes.data_as(i32p)
        st.edge_b = self.edge_b.ctypes.data_as(i32p)
        st.edge_meas = self.edge_meas.ctypes.data_as(f64p)
        return st

    def n_free(self):
        return int((self.node_fixed == 0).sum())


def se3_log(pose7):
    pose7 = np.ascontiguousarray(pose7, np.float64)
    out = np.zeros(6)
    lib().orc_se3_log(pose7.ctypes.data_as(f64p), out.ctypes.data_as(f64p))
    return out


def pgo_residual_jacobian(pose_c, pose_n, meas):
    pc, pn = np.ascontiguousarray(pose_c, np.float64), np.ascontiguousarray(pose_n, np.float64)
    m = np.ascontiguousarray(meas, np.float64)
    r, Jc, Jn = np.zeros(6), np.zeros((6, 6)), np.zeros((6, 6))
    lib().orc_pgo_residual_jacobian(pc.ctypes.data_as(f64p), pn.ctypes.data_as(f64p), m.ctypes.data_as(f64p),
                                    r.ctypes.data_as(f64p), Jc.ctypes.data_as(f64p), Jn.ctypes.data_as(f64p))
    return r, Jc, Jn


def pgo_linearize(arr, use_huber=True, huber=1.0):
    st = arr.fill(PgoProblem())
    o = _opts(use_huber, huber, 0, 0, 1)
    n = 6 * arr.n_free()
    H, g = np.zeros((max(n, 1), max(n, 1))), np.zeros(max(n, 1))
    cost, nf = C.c_double(), C.c_int32()
    lib().orc_pgo_linearize(C.byref(st), C.byref(o), H.ctypes.data_as(f64p), g.ctypes.data_as(f64p), C.byref(cost), C.byref(nf))
    return H[:n, :n].copy(), g[:n].copy(), cost.value


def pose_graph_optimize(arr, use_huber=True, huber=1.0, max_iters=20):
    st = arr.fill(PgoProblem())
    o = _opts(use_huber, huber, max_iters, 0, 1)
    s = BaSummary()
    lib().orc_pose_graph_optimize(C.byref(st), C.byref(o), C.byref(s))
    return s
