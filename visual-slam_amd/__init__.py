"""visual_slam_amd -- Python plumbing over the C ABI of libvslam_hip.so (include/vslam_hip.h).

The product is the shared library (hand-written HIP for gfx950 + C++ host code); this module only
loads it with ctypes so that tests/ and bench.py can drive the C ABI.  There is NO fallback: if the
library is missing or no MI355X is visible, `load()` / `Context()` raise.

The directory is named `visual-slam_amd` (the layout the build contract asks for); it is imported
under the module name `visual_slam_amd` -- see `__graft_entry__.load_package()`.
"""
import ctypes as C
import os
from pathlib import Path

import numpy as np

_DIR = Path(__file__).resolve().parent
_SO = _DIR / "libvslam_hip.so"
_LIB = None

u8p = C.POINTER(C.c_uint8)
i32p = C.POINTER(C.c_int32)
u32p = C.POINTER(C.c_uint32)
u64p = C.POINTER(C.c_uint64)
i64p = C.POINTER(C.c_int64)
f32p = C.POINTER(C.c_float)
f64p = C.POINTER(C.c_double)

STAGES = ["response", "select", "describe", "match", "match_finalize", "ba_linearize", "ba_schur",
          "ba_solve", "bow_transform", "bow_score", "ba_finish", "ba_step"]

OK = 0
ERR = {-1: "VSL_ERR_INVALID", -2: "VSL_ERR_HIP", -3: "VSL_ERR_NOMEM", -4: "VSL_ERR_CAPACITY",
       -5: "VSL_ERR_NO_DEVICE", -6: "VSL_ERR_IO", -7: "VSL_ERR_NUMERIC"}


class VslError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("%s (%d): %s" % (ERR.get(code, "VSL_ERR_?"), code, msg))
        self.code = code


class BaProblem(C.Structure):
    _fields_ = [("n_cams", C.c_int32), ("n_lms", C.c_int32), ("n_obs", C.c_int32),
                ("cam_model", C.c_int32 * 2), ("poses", f64p), ("cam_fixed", u8p),
                ("cam_intr", i32p), ("intr", f64p), ("points", f64p), ("obs_cam", i32p),
                ("obs_lm", i32p), ("obs_uv", f64p)]


class BaOptions(C.Structure):
    _fields_ = [("use_huber", C.c_int32), ("huber_parameter", C.c_double),
                ("max_num_iterations", C.c_int32), ("verbosity", C.c_int32)]


class BaSummary(C.Structure):
    _fields_ = [("initial_cost", C.c_double), ("final_cost", C.c_double),
                ("iterations", C.c_int32), ("successful_steps", C.c_int32),
                ("termination", C.c_int32), ("linearize_ms", C.c_double),
                ("schur_ms", C.c_double), ("solve_ms", C.c_double), ("total_ms", C.c_double)]


class PgoProblem(C.Structure):
    _fields_ = [("n_nodes", C.c_int32), ("n_edges", C.c_int32), ("poses", f64p), ("node_fixed", u8p),
                ("edge_a", i32p), ("edge_b", i32p), ("edge_meas", f64p)]


def library_path():
    return _SO


def load():
    """dlopen libvslam_hip.so (raises if it has not been built)."""
    global _LIB
    if _LIB is None:
        if not _SO.exists():
            raise ImportError("%s is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                              "(there is no CPU fallback)" % _SO)
        try:
            # PyTorch-ROCm ships its own HIP runtime; when both live in one process the runtime that is
            # loaded FIRST must be torch's, otherwise torch later reports "No HIP GPUs are available".
            import torch  # noqa: F401
        except ImportError:
            pass
        L = C.CDLL(str(_SO))
        L.vsl_version.restype = C.c_char_p
        L.vsl_last_error.restype = C.c_char_p
        L.vsl_last_error.argtypes = [C.c_void_p]
        L.vsl_ctx_stream.restype = C.c_void_p
        L.vsl_ctx_stream.argtypes = [C.c_void_p]
        L.vsl_frames_images_dev.restype = C.c_void_p
        L.vsl_frames_images_dev.argtypes = [C.c_void_p]
        _LIB = L
    return _LIB


def device_count():
    return load().vsl_device_count()


def _img(img):
    img = np.ascontiguousarray(img, dtype=np.uint8)
    assert img.ndim == 2
    return img, img.ctypes.data_as(u8p), img.shape[1], img.shape[0], C.c_size_t(img.strides[0])


class Context:
    """vsl_ctx: one HIP stream + scratch.  `stream` = an existing hipStream_t handle (int) to borrow."""

    def __init__(self, device=0, stream=None):
        self.L = load()
        h = C.c_void_p()
        if stream is None:
            rc = self.L.vsl_ctx_create(int(device), C.byref(h))
        else:
            rc = self.L.vsl_ctx_create_on_stream(int(device), C.c_void_p(stream), C.byref(h))
        if rc != OK:
            raise VslError(rc, self.L.vsl_last_error(None).decode())
        self.h = h

    def close(self):
        if getattr(self, "h", None):
            self.L.vsl_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _ck(self, rc):
        if rc != OK:
            raise VslError(rc, self.L.vsl_last_error(self.h).decode())

    def synchronize(self):
        self._ck(self.L.vsl_ctx_synchronize(self.h))

    def stream(self):
        return self.L.vsl_ctx_stream(self.h)

    def wait_event(self, ev):
        """Device-side ordering: later work on this context starts after the work `ev` marked has completed."""
        self._ck(self.L.vsl_ctx_wait_event(self.h, ev.h))

    def set_profiling(self, on):
        self._ck(self.L.vsl_ctx_set_profiling(self.h, int(on)))

    def last_ba_layout(self):
        """(doubles of S, band form, bandwidth) of the last general-path bundle adjustment set up on this context; band form
        0 = dense, 1 = band (cameras in reverse Cuthill-McKee order), 2 = cyclic band (cameras as they came, the band
        closes around the loop)."""
        e, b, w = C.c_int64(), C.c_int(), C.c_int()
        self._ck(self.L.vsl_ctx_last_ba_layout(self.h, C.byref(e), C.byref(b), C.byref(w)))
        return e.value, b.value, w.value

    def reset_profiling(self):
        self._ck(self.L.vsl_ctx_reset_profiling(self.h))

    def stage_ms(self):
        out = {}
        for i, name in enumerate(STAGES):
            ms, n = C.c_double(), C.c_int64()
            self._ck(self.L.vsl_ctx_stage_ms(self.h, i, C.byref(ms), C.byref(n)))
            out[name] = (ms.value, n.value)
        return out

    def set_tie_eps(self, eps):
        self._ck(self.L.vsl_ctx_set_tie_eps(self.h, C.c_double(eps)))

    def sqrt_check(self, lo_bits, hi_bits):
        """Mismatches between the response kernel's square root and sqrtf over the float bit patterns [lo, hi]."""
        n = C.c_ulonglong(0)
        self._ck(self.L.vsl_diag_sqrt_check(self.h, C.c_uint32(int(lo_bits)), C.c_uint32(int(hi_bits)), C.byref(n)))
        return int(n.value)

    def set_diagnostic(self, name, value):
        self._ck(self.L.vsl_ctx_set_diagnostic(self.h, name.encode(), C.c_int(int(value))))

    def spd_solve(self, S, b, half_bandwidth=-1):
        """Solves S x = b with the reduced-camera-system solver (dense, or band storage when half_bandwidth >= 0)."""
        S = np.ascontiguousarray(S, np.float64)
        b = np.ascontiguousarray(b, np.float64)
        n = len(b)
        assert S.shape == (n, n)
        x = np.zeros(n, np.float64)
        self._ck(self.L.vsl_spd_solve(self.h, S.ctypes.data_as(f64p), b.ctypes.data_as(f64p), n, int(half_bandwidth),
                                      x.ctypes.data_as(f64p)))
        return x

    def spd_solve_cyclic(self, S, b, half_bandwidth):
        """Solves S x = b for a CYCLIC band matrix (non-zeros within cyclic distance half_bandwidth of the diagonal)."""
        S = np.ascontiguousarray(S, np.float64)
        b = np.ascontiguousarray(b, np.float64)
        n = len(b)
        assert S.shape == (n, n)
        x = np.zeros(n, np.float64)
        self._ck(self.L.vsl_spd_solve_cyclic(self.h, S.ctypes.data_as(f64p), b.ctypes.data_as(f64p), n, int(half_bandwidth),
                                             x.ctypes.data_as(f64p)))
        return x

    # ---- keypoints.h drop-ins (host buffers)
    def detect_describe(self, img, num_features=1500, rotate=True):
        img, p, w, h, pitch = _img(img)
        cap = num_features
        xy = np.zeros((cap, 2), np.float64)
        ang = np.zeros(cap, np.float64)
        desc = np.zeros((cap, 4), np.uint64)
        n = C.c_int32()
        self._ck(self.L.vsl_detect_describe(self.h, p, w, h, pitch, int(num_features), int(rotate), cap,
                                            xy.ctypes.data_as(f64p), ang.ctypes.data_as(f64p),
                                            desc.ctypes.data_as(u64p), C.byref(n)))
        return xy[:n.value].copy(), ang[:n.value].copy(), desc[:n.value].copy()

    def detect_keypoints(self, img, num_features=1500):
        img, p, w, h, pitch = _img(img)
        xy = np.zeros((num_features, 2), np.float64)
        n = C.c_int32()
        self._ck(self.L.vsl_detect_keypoints(self.h, p, w, h, pitch, int(num_features), num_features,
                                             xy.ctypes.data_as(f64p), C.byref(n)))
        return xy[:n.value].copy()

    def compute_angles(self, img, corners, rotate=True):
        img, p, w, h, pitch = _img(img)
        corners = np.ascontiguousarray(corners, np.float64).reshape(-1, 2)
        ang = np.zeros(len(corners), np.float64)
        self._ck(self.L.vsl_compute_angles(self.h, p, w, h, pitch, corners.ctypes.data_as(f64p),
                                           len(corners), int(rotate), ang.ctypes.data_as(f64p)))
        return ang

    def compute_descriptors(self, img, corners, angles):
        img, p, w, h, pitch = _img(img)
        corners = np.ascontiguousarray(corners, np.float64).reshape(-1, 2)
        angles = np.ascontiguousarray(angles, np.float64)
        desc = np.zeros((len(corners), 4), np.uint64)
        self._ck(self.L.vsl_compute_descriptors(self.h, p, w, h, pitch, corners.ctypes.data_as(f64p),
                                                angles.ctypes.data_as(f64p), len(corners),
                                                desc.ctypes.data_as(u64p)))
        return desc

    def min_eig_response(self, img):
        img, p, w, h, pitch = _img(img)
        out = np.zeros((h, w), np.float32)
        self._ck(self.L.vsl_min_eig_response(self.h, p, w, h, pitch, out.ctypes.data_as(f32p)))
        return out

    def match_descriptors(self, d1, d2, threshold=70, dist_2_best=1.2):
        d1 = np.ascontiguousarray(d1, np.uint64).reshape(-1, 4)
        d2 = np.ascontiguousarray(d2, np.uint64).reshape(-1, 4)
        pairs = np.zeros((max(1, min(len(d1), len(d2))), 2), np.int32)
        n = C.c_int32()
        self._ck(self.L.vsl_match_descriptors(self.h, d1.ctypes.data_as(u64p), len(d1),
                                              d2.ctypes.data_as(u64p), len(d2), int(threshold),
                                              C.c_double(dist_2_best), pairs.ctypes.data_as(i32p),
                                              C.byref(n)))
        return pairs[:n.value].copy()

    # ---- vo_utils.h drop-ins
    def project_landmarks(self, pose7, model, intr8, width, height, points, cam_z_threshold=0.1):
        pose7 = np.ascontiguousarray(pose7, np.float64)
        intr8 = np.ascontiguousarray(intr8, np.float64)
        points = np.ascontiguousarray(points, np.float64).reshape(-1, 3)
        n = len(points)
        uv = np.zeros((max(n, 1), 2), np.float64)
        idx = np.zeros(max(n, 1), np.int32)
        m = C.c_int32()
        self._ck(self.L.vsl_project_landmarks(self.h, pose7.ctypes.data_as(f64p), int(model), intr8.ctypes.data_as(f64p),
                                              int(width), int(height), points.ctypes.data_as(f64p), n,
                                              C.c_double(cam_z_threshold), uv.ctypes.data_as(f64p),
                                              idx.ctypes.data_as(i32p), C.byref(m)))
        return uv[:m.value].copy(), idx[:m.value].copy()

    def find_matches_landmarks(self, kp_xy, kp_desc, proj_uv, proj_lm, lm_obs_start, obs_desc, max_dist_2d=20.0,
                               threshold=70, dist_2_best=1.2):
        kp_xy = np.ascontiguousarray(kp_xy, np.float64).reshape(-1, 2)
        kp_desc = np.ascontiguousarray(kp_desc, np.uint64).reshape(-1, 4)
        proj_uv = np.ascontiguousarray(proj_uv, np.float64).reshape(-1, 2)
        proj_lm = np.ascontiguousarray(proj_lm, np.int32)
        lm_obs_start = np.ascontiguousarray(lm_obs_start, np.int32)
        obs_desc = np.ascontiguousarray(obs_desc, np.uint64).reshape(-1, 4)
        pairs = np.zeros((max(len(kp_xy), 1), 2), np.int32)
        m = C.c_int32()
        self._ck(self.L.vsl_find_matches_landmarks(self.h, kp_xy.ctypes.data_as(f64p), kp_desc.ctypes.data_as(u64p),
                                                   len(kp_xy), proj_uv.ctypes.data_as(f64p),
                                                   proj_lm.ctypes.data_as(i32p), len(proj_uv),
                                                   lm_obs_start.ctypes.data_as(i32p), len(lm_obs_start) - 1,
                                                   obs_desc.ctypes.data_as(u64p), C.c_double(max_dist_2d),
                                                   int(threshold), C.c_double(dist_2_best),
                                                   pairs.ctypes.data_as(i32p), C.byref(m)))
        return pairs[:m.value].copy()

    def orb_detect_describe(self, img, num_features=1500):
        """ORB front end of compute_bow_vector: (kp[n, 5] = x, y, angle_deg, response, octave; desc[n, 32])."""
        img, p, w, h, pitch = _img(img)
        cap = 2 * num_features + 512
        kp = np.zeros((cap, 5), np.float32)
        desc = np.zeros((cap, 32), np.uint8)
        n = C.c_int32()
        self._ck(self.L.vsl_orb_detect_describe(self.h, p, w, h, pitch, int(num_features), cap,
                                                kp.ctypes.data_as(C.POINTER(C.c_float)), desc.ctypes.data_as(u8p), C.byref(n)))
        return kp[:n.value].copy(), desc[:n.value].copy()

    # ---- bundle adjustment
    def _ba_struct(self, arr):
        st = BaProblem()
        st.n_cams, st.n_lms, st.n_obs = len(arr.poses), len(arr.points), len(arr.obs_cam)
        st.cam_model[0], st.cam_model[1] = arr.cam_model
        st.poses = arr.poses.ctypes.data_as(f64p)
        st.cam_fixed = arr.cam_fixed.ctypes.data_as(u8p)
        st.cam_intr = arr.cam_intr.ctypes.data_as(i32p)
        st.intr = arr.intr.ctypes.data_as(f64p)
        st.points = arr.points.ctypes.data_as(f64p)
        st.obs_cam = arr.obs_cam.ctypes.data_as(i32p)
        st.obs_lm = arr.obs_lm.ctypes.data_as(i32p)
        st.obs_uv = arr.obs_uv.ctypes.data_as(f64p)
        return st

    @staticmethod
    def _ba_opts(use_huber, huber, max_iters, verbosity):
        o = BaOptions()
        o.use_huber, o.huber_parameter = int(use_huber), float(huber)
        o.max_num_iterations, o.verbosity = int(max_iters), int(verbosity)
        return o

    def bundle_adjust(self, arr, use_huber=True, huber=1.0, max_iters=20, verbosity=0):
        """arr: object with the numpy fields of include/vslam_hip.h's vsl_ba_problem (optimised in place)."""
        st = self._ba_struct(arr)
        o = self._ba_opts(use_huber, huber, max_iters, verbosity)
        s = BaSummary()
        self._ck(self.L.vsl_bundle_adjust(self.h, C.byref(st), C.byref(o), C.byref(s)))
        return s

    # ---- pose graph optimisation (loop_closure_utils.h:446-587)
    @staticmethod
    def _pgo_struct(arr):
        st = PgoProblem()
        st.n_nodes, st.n_edges = len(arr.poses), len(arr.edge_a)
        st.poses = arr.poses.ctypes.data_as(f64p)
        st.node_fixed = arr.node_fixed.ctypes.data_as(u8p)
        st.edge_a = arr.edge_a.ctypes.data_as(i32p)
        st.edge_b = arr.edge_b.ctypes.data_as(i32p)
        st.edge_meas = arr.edge_meas.ctypes.data_as(f64p)
        return st

    def pose_graph_optimize(self, arr, use_huber=True, huber=1.0, max_iters=20, verbosity=0):
        """arr: object with the numpy fields of vsl_pgo_problem (poses [N, 7] optimised in place)."""
        st = self._pgo_struct(arr)
        o = self._ba_opts(use_huber, huber, max_iters, verbosity)
        s = BaSummary()
        self._ck(self.L.vsl_pose_graph_optimize(self.h, C.byref(st), C.byref(o), C.byref(s)))
        return s

    def pgo_linearize(self, arr, use_huber=True, huber=1.0):
        st = self._pgo_struct(arr)
        o = self._ba_opts(use_huber, huber, 0, 0)
        n = 6 * int((arr.node_fixed == 0).sum())
        H, g = np.zeros((max(n, 1), max(n, 1))), np.zeros(max(n, 1))
        cost, nf = C.c_double(), C.c_int32()
        Hc = np.zeros(n * n)
        self._ck(self.L.vsl_pgo_linearize(self.h, C.byref(st), C.byref(o), Hc.ctypes.data_as(f64p), g.ctypes.data_as(f64p),
                                          C.byref(cost), C.byref(nf)))
        return Hc.reshape(n, n), g[:n].copy(), cost.value

    def bundle_adjust_intrinsics(self, arr, use_huber=True, huber=1.0, max_iters=20, verbosity=0):
        """optimize_intrinsics = true: optimises arr.poses / arr.points / arr.intr in place."""
        st = self._ba_struct(arr)
        o = self._ba_opts(use_huber, huber, max_iters, verbosity)
        out = BaSummary()
        self._ck(self.L.vsl_bundle_adjust_intrinsics(self.h, C.byref(st), C.byref(o), arr.intr.ctypes.data_as(f64p),
                                                     C.byref(out)))
        return out

    def ba_linearize(self, arr, use_huber=True, huber=1.0, lm_first=0, lm_count=-1):
        st = self._ba_struct(arr)
        o = self._ba_opts(use_huber, huber, 0, 0)
        n = 6 * int((arr.cam_fixed == 0).sum())
        S = np.zeros((n, n))
        g = np.zeros(n)
        cost, nf = C.c_double(), C.c_int32()
        self._ck(self.L.vsl_ba_linearize(self.h, C.byref(st), C.byref(o), int(lm_first), int(lm_count),
                                         S.ctypes.data_as(f64p), g.ctypes.data_as(f64p), C.byref(cost),
                                         C.byref(nf)))
        return S, g, cost.value

    def ba_residuals_jacobians(self, arr):
        st = self._ba_struct(arr)
        n = len(arr.obs_cam)
        r, Jp, Jl = np.zeros((n, 2)), np.zeros((n, 2, 6)), np.zeros((n, 2, 3))
        self._ck(self.L.vsl_ba_residuals_jacobians(self.h, C.byref(st), r.ctypes.data_as(f64p),
                                                   Jp.ctypes.data_as(f64p), Jl.ctypes.data_as(f64p)))
        return r, Jp, Jl

    # ---- DBoW2
    def load_vocabulary(self, path):
        return Vocabulary(self, path)

    def bow_score_batch(self, q_ids, q_vals, cands):
        """cands: list of (ids, vals).  Returns the L1 scores (float64 array)."""
        q_ids = np.ascontiguousarray(q_ids, np.uint32)
        q_vals = np.ascontiguousarray(q_vals, np.float64)
        m = len(cands)
        offs = np.zeros(m + 1, np.int32)
        for i, (ids, _) in enumerate(cands):
            offs[i + 1] = offs[i] + len(ids)
        c_ids = np.concatenate([np.asarray(c[0], np.uint32) for c in cands]) if m else np.zeros(0, np.uint32)
        c_vals = np.concatenate([np.asarray(c[1], np.float64) for c in cands]) if m else np.zeros(0)
        c_ids = np.ascontiguousarray(c_ids, np.uint32)
        c_vals = np.ascontiguousarray(c_vals, np.float64)
        scores = np.zeros(max(m, 1), np.float64)
        self._ck(self.L.vsl_bow_score_batch(self.h, q_ids.ctypes.data_as(u32p), q_vals.ctypes.data_as(f64p),
                                            len(q_ids), c_ids.ctypes.data_as(u32p),
                                            c_vals.ctypes.data_as(f64p), offs.ctypes.data_as(i32p), m,
                                            scores.ctypes.data_as(f64p)))
        return scores[:m].copy()


class BaArrays:
    """Owns the contiguous numpy arrays of a flattened bundle-adjustment problem (the layout of vsl_ba_problem,
    include/vslam_hip.h): what Context.bundle_adjust / ba_linearize / BaSession take."""

    def __init__(self, poses, cam_fixed, cam_intr, intr, points, obs_cam, obs_lm, obs_uv, cam_model=(0, 0)):
        self.poses = np.ascontiguousarray(poses, np.float64).reshape(-1, 7).copy()
        self.cam_fixed = np.ascontiguousarray(cam_fixed, np.uint8).copy()
        self.cam_intr = np.ascontiguousarray(cam_intr, np.int32).copy()
        self.intr = np.ascontiguousarray(intr, np.float64).reshape(2, 8).copy()
        self.points = np.ascontiguousarray(points, np.float64).reshape(-1, 3).copy()
        self.obs_cam = np.ascontiguousarray(obs_cam, np.int32).copy()
        self.obs_lm = np.ascontiguousarray(obs_lm, np.int32).copy()
        self.obs_uv = np.ascontiguousarray(obs_uv, np.float64).reshape(-1, 2).copy()
        self.cam_model = tuple(int(m) for m in cam_model)

    @classmethod
    def from_dict(cls, d):
        return cls(d["poses"], d["cam_fixed"], d["cam_intr"], d["intr"], d["points"], d["obs_cam"], d["obs_lm"], d["obs_uv"],
                   d.get("cam_model", (0, 0)))

    def copy(self):
        return BaArrays(self.poses, self.cam_fixed, self.cam_intr, self.intr, self.points, self.obs_cam, self.obs_lm,
                        self.obs_uv, self.cam_model)

    @property
    def n_free(self):
        return int((self.cam_fixed == 0).sum())


class BaSession:
    """vsl_ba_session: the step-wise (multi-GPU) bundle-adjustment API; see ba_dist.py for the loop."""

    def __init__(self, ctx, arr, use_huber=True, huber=1.0, lm_first=0, lm_count=None):
        self.ctx = ctx
        st = ctx._ba_struct(arr)
        o = ctx._ba_opts(use_huber, huber, 0, 0)
        if lm_count is None:
            lm_count = len(arr.points) - lm_first
        h = C.c_void_p()
        ctx._ck(ctx.L.vsl_ba_session_create(ctx.h, C.byref(st), C.byref(o), int(lm_first), int(lm_count), C.byref(h)))
        self.h = h
        v = [C.c_int32() for _ in range(4)]
        ctx._ck(ctx.L.vsl_ba_session_dims(self.h, *[C.byref(x) for x in v]))
        self.n, self.n_lms, self.n_obs, self.n_cams = (x.value for x in v)
        self.lm_first, self.lm_count = int(lm_first), int(lm_count)

    def close(self):
        if getattr(self, "h", None):
            self.ctx.L.vsl_ba_session_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def linearize(self, use_scale):
        self.ctx._ck(self.ctx.L.vsl_ba_session_linearize(self.h, int(use_scale)))

    def hdiag_cost(self, out_ptr):
        self.ctx._ck(self.ctx.L.vsl_ba_session_hdiag_cost_dev(self.h, C.c_void_p(out_ptr)))

    def set_scale(self, hdiag_ptr):
        self.ctx._ck(self.ctx.L.vsl_ba_session_set_scale_dev(self.h, C.c_void_p(hdiag_ptr)))

    def reduce(self, radius, packb_ptr, gmax_ptr):
        self.ctx._ck(self.ctx.L.vsl_ba_session_reduce_dev(self.h, C.c_double(radius), C.c_void_p(packb_ptr),
                                                          C.c_void_p(gmax_ptr)))

    def step(self, packb_ptr, radius, refresh, packc_ptr):
        self.ctx._ck(self.ctx.L.vsl_ba_session_step_dev(self.h, C.c_void_p(packb_ptr), C.c_double(radius),
                                                        int(refresh), C.c_void_p(packc_ptr)))

    def accept(self):
        self.ctx._ck(self.ctx.L.vsl_ba_session_accept(self.h))

    def download(self):
        poses = np.zeros((self.n_cams, 7))
        pts = np.zeros((self.n_lms, 3))
        self.ctx._ck(self.ctx.L.vsl_ba_session_download(self.h, poses.ctypes.data_as(f64p), pts.ctypes.data_as(f64p)))
        return poses, pts


class Vocabulary:
    def __init__(self, ctx, path):
        self.ctx = ctx
        h = C.c_void_p()
        ctx._ck(ctx.L.vsl_voc_load_text(ctx.h, os.fsencode(str(path)), C.byref(h)))
        self.h = h

    def info(self):
        v = [C.c_int32() for _ in range(4)]
        self.ctx._ck(self.ctx.L.vsl_voc_info(self.h, *[C.byref(x) for x in v]))
        return tuple(x.value for x in v)

    def transform(self, desc32, levelsup=4):
        desc32 = np.ascontiguousarray(desc32, np.uint8).reshape(-1, 32)
        n = len(desc32)
        ids = np.zeros(max(n, 1), np.uint32)
        vals = np.zeros(max(n, 1), np.float64)
        fn = np.zeros(max(n, 1), np.uint32)
        ff = np.zeros(max(n, 1), np.uint32)
        nnz, fvn = C.c_int32(), C.c_int32()
        self.ctx._ck(self.ctx.L.vsl_bow_transform(self.ctx.h, self.h, desc32.ctypes.data_as(u8p), n,
                                                  int(levelsup), ids.ctypes.data_as(u32p),
                                                  vals.ctypes.data_as(f64p), C.byref(nnz),
                                                  fn.ctypes.data_as(u32p), ff.ctypes.data_as(u32p),
                                                  C.byref(fvn)))
        return (ids[:nnz.value].copy(), vals[:nnz.value].copy(), fn[:fvn.value].copy(),
                ff[:fvn.value].copy())

    def compute_bow_vector(self, img, num_features=1500, levelsup=4):
        """compute_bow_vector (keypoints.h:243-254): ORB front end + transform, one call."""
        img, p, w, h, pitch = _img(img)
        cap = 2 * num_features + 512
        ids = np.zeros(cap, np.uint32)
        vals = np.zeros(cap, np.float64)
        fn = np.zeros(cap, np.uint32)
        ff = np.zeros(cap, np.uint32)
        nnz, fvn = C.c_int32(), C.c_int32()
        self.ctx._ck(self.ctx.L.vsl_compute_bow_vector(self.ctx.h, self.h, p, w, h, pitch, int(num_features), int(levelsup), cap,
                                                       ids.ctypes.data_as(u32p), vals.ctypes.data_as(f64p), C.byref(nnz),
                                                       fn.ctypes.data_as(u32p), ff.ctypes.data_as(u32p), C.byref(fvn)))
        return (ids[:nnz.value].copy(), vals[:nnz.value].copy(), fn[:fvn.value].copy(), ff[:fvn.value].copy())

    def close(self):
        if getattr(self, "h", None):
            self.ctx.L.vsl_voc_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class BowDatabase:
    """Device-resident BowVectors (vsl_bowdb_*): append once per keyframe, score a query against any subset."""

    def __init__(self, ctx, cap_entries=1 << 20, cap_vectors=1024):
        self.ctx = ctx
        h = C.c_void_p()
        ctx._ck(ctx.L.vsl_bowdb_create(ctx.h, C.c_int64(int(cap_entries)), int(cap_vectors), C.byref(h)))
        self.h = h

    def append(self, ids, vals):
        ids = np.ascontiguousarray(ids, np.uint32)
        vals = np.ascontiguousarray(vals, np.float64)
        assert len(ids) == len(vals)
        idx = C.c_int32()
        self.ctx._ck(self.ctx.L.vsl_bowdb_append(self.ctx.h, self.h, ids.ctypes.data_as(u32p), vals.ctypes.data_as(f64p),
                                                 len(ids), C.byref(idx)))
        return idx.value

    def info(self):
        n, e = C.c_int32(), C.c_int64()
        self.ctx._ck(self.ctx.L.vsl_bowdb_info(self.h, C.byref(n), C.byref(e)))
        return n.value, e.value

    def score(self, q_ids, q_vals, cand_index=None, m=None):
        """L1 scores of the query against vectors cand_index (or the first m / all vectors)."""
        q_ids = np.ascontiguousarray(q_ids, np.uint32)
        q_vals = np.ascontiguousarray(q_vals, np.float64)
        if cand_index is not None:
            cand_index = np.ascontiguousarray(cand_index, np.int32)
            m = len(cand_index)
            ip = cand_index.ctypes.data_as(i32p)
        else:
            m = self.info()[0] if m is None else int(m)
            ip = None
        scores = np.zeros(max(m, 1), np.float64)
        self.ctx._ck(self.ctx.L.vsl_bowdb_score(self.ctx.h, self.h, q_ids.ctypes.data_as(u32p), q_vals.ctypes.data_as(f64p),
                                                len(q_ids), ip, m, scores.ctypes.data_as(f64p)))
        return scores[:m].copy()

    def close(self):
        if getattr(self, "h", None):
            self.ctx.L.vsl_bowdb_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Event:
    """vsl_event: marks a point in one context's stream for another context to wait on (device side)."""

    def __init__(self, ctx):
        self.L = ctx.L
        h = C.c_void_p()
        ctx._ck(self.L.vsl_event_create(ctx.h, C.byref(h)))
        self.h = h

    def record(self, ctx):
        ctx._ck(self.L.vsl_event_record(self.h, ctx.h))

    def close(self):
        if getattr(self, "h", None):
            self.L.vsl_event_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Frames:
    """vsl_frames: device-resident batched frame store."""

    def __init__(self, ctx, max_images, w, h, max_features=1500, max_pairs=None):
        self.ctx = ctx
        self.max_images, self.w, self.hgt, self.F = max_images, w, h, max_features
        self.max_pairs = max_pairs if max_pairs is not None else max(1, max_images // 2)
        hnd = C.c_void_p()
        ctx._ck(ctx.L.vsl_frames_create(ctx.h, int(max_images), int(w), int(h), int(max_features),
                                        int(self.max_pairs), C.byref(hnd)))
        self.h = hnd

    def close(self):
        if getattr(self, "h", None):
            self.ctx.L.vsl_frames_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def images_dev(self):
        return self.ctx.L.vsl_frames_images_dev(self.h)

    def upload(self, first, imgs):
        imgs = np.ascontiguousarray(imgs, np.uint8)
        if imgs.ndim == 2:
            imgs = imgs[None]
        n, h, w = imgs.shape
        assert (h, w) == (self.hgt, self.w)
        self.ctx._ck(self.ctx.L.vsl_frames_upload(self.ctx.h, self.h, int(first), n,
                                                  imgs.ctypes.data_as(u8p), C.c_size_t(w),
                                                  C.c_size_t(w * h)))
        self.ctx.synchronize()

    def upload_async(self, first, imgs, ctx=None):
        """Enqueue the copy of a dense (n, h, w) uint8 batch -- pinned host memory for a truly asynchronous copy --
        on `ctx` (default: the store's own context) and return without waiting; the caller keeps `imgs` alive and
        unchanged until that stream has passed the copy."""
        c = ctx or self.ctx
        assert imgs.dtype == np.uint8 and imgs.ndim == 3 and imgs.flags["C_CONTIGUOUS"]
        n, h, w = imgs.shape
        assert (h, w) == (self.hgt, self.w)
        c._ck(c.L.vsl_frames_upload(c.h, self.h, int(first), n, imgs.ctypes.data_as(u8p), C.c_size_t(w),
                                    C.c_size_t(w * h)))

    def detect_describe(self, first, n, num_features=1500, rotate=True):
        self.ctx._ck(self.ctx.L.vsl_frames_detect_describe(self.ctx.h, self.h, int(first), int(n),
                                                           int(num_features), int(rotate)))

    def resolve_ties(self):
        n = C.c_int32()
        self.ctx._ck(self.ctx.L.vsl_frames_resolve_ties(self.ctx.h, self.h, C.byref(n)))
        return n.value

    def exact_fallbacks(self):
        return int(self.ctx.L.vsl_frames_exact_fallbacks(self.h))

    def match(self, slot_pairs, threshold=70, dist_2_best=1.2):
        sp = np.ascontiguousarray(slot_pairs, np.int32).reshape(-1, 2)
        self.ctx._ck(self.ctx.L.vsl_frames_match(self.ctx.h, self.h, sp.ctypes.data_as(i32p), len(sp),
                                                 int(threshold), C.c_double(dist_2_best)))

    def keypoints(self, slot):
        xy = np.zeros((self.F, 2), np.float64)
        ang = np.zeros(self.F, np.float64)
        desc = np.zeros((self.F, 4), np.uint64)
        n = C.c_int32()
        self.ctx._ck(self.ctx.L.vsl_frames_download_keypoints(self.ctx.h, self.h, int(slot), self.F,
                                                              xy.ctypes.data_as(f64p),
                                                              ang.ctypes.data_as(f64p),
                                                              desc.ctypes.data_as(u64p), C.byref(n)))
        return xy[:n.value].copy(), ang[:n.value].copy(), desc[:n.value].copy()

    def matches(self, pair):
        out = np.zeros((self.F, 2), np.int32)
        n = C.c_int32()
        self.ctx._ck(self.ctx.L.vsl_frames_download_matches(self.ctx.h, self.h, int(pair), self.F,
                                                            out.ctypes.data_as(i32p), C.byref(n)))
        return out[:n.value].copy()

    def candidate_counts(self, n_images):
        nc = np.zeros(max(n_images, 1), np.int32)
        self.ctx._ck(self.ctx.L.vsl_frames_download_candidate_counts(self.ctx.h, self.h, int(n_images),
                                                                     nc.ctypes.data_as(i32p)))
        return nc[:n_images].copy()

    def counts(self, n_images, n_pairs):
        nk = np.zeros(max(n_images, 1), np.int32)
        nm = np.zeros(max(n_pairs, 1), np.int32)
        self.ctx._ck(self.ctx.L.vsl_frames_download_counts(self.ctx.h, self.h, int(n_images),
                                                           nk.ctypes.data_as(i32p), int(n_pairs),
                                                           nm.ctypes.data_as(i32p)))
        return nk[:n_images].copy(), nm[:n_pairs].copy()


class Map:
    """vsl_map: device-resident landmark table + observation-descriptor pool (per-frame tracking)."""

    def __init__(self, ctx, cap_landmarks=4096, cap_descriptors=16384):
        self.ctx = ctx
        hnd = C.c_void_p()
        ctx._ck(ctx.L.vsl_map_create(ctx.h, int(cap_landmarks), int(cap_descriptors), C.byref(hnd)))
        self.h = hnd

    def close(self):
        if getattr(self, "h", None):
            self.ctx.L.vsl_map_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def append_descriptors(self, desc):
        desc = np.ascontiguousarray(desc, np.uint64).reshape(-1, 4)
        first = C.c_int32()
        self.ctx._ck(self.ctx.L.vsl_map_append_descriptors(self.h, len(desc), desc.ctypes.data_as(u64p), C.byref(first)))
        return first.value

    def append_descriptors_from_frame(self, frames, slot, feature_ids):
        ids = np.ascontiguousarray(feature_ids, np.int32)
        first = C.c_int32()
        self.ctx._ck(self.ctx.L.vsl_map_append_descriptors_from_frame(self.h, frames.h, int(slot), len(ids),
                                                                      ids.ctypes.data_as(i32p), C.byref(first)))
        return first.value

    def set_landmarks(self, points, obs_start, obs_pool_index):
        points = np.ascontiguousarray(points, np.float64).reshape(-1, 3)
        obs_start = np.ascontiguousarray(obs_start, np.int32)
        obs_pool_index = np.ascontiguousarray(obs_pool_index, np.int32)
        self.ctx._ck(self.ctx.L.vsl_map_set_landmarks(self.h, len(points), points.ctypes.data_as(f64p),
                                                      obs_start.ctypes.data_as(i32p), obs_pool_index.ctypes.data_as(i32p)))

    def info(self):
        a, b, c = C.c_int32(), C.c_int32(), C.c_int32()
        self.ctx._ck(self.ctx.L.vsl_map_info(self.h, C.byref(a), C.byref(b), C.byref(c)))
        return a.value, b.value, c.value

    def track(self, frames, slot, pose7, model, intr8, width, height, cam_z_threshold=0.1, max_dist_2d=20.0, threshold=70,
              dist_2_best=1.2, with_corners=False):
        """project_landmarks + find_matches_landmarks against the frame store slot: (pairs, n_projected); with_corners:
        also the slot's keypoint positions from the same round trip (vsl_map_track_corners)."""
        pose7 = np.ascontiguousarray(pose7, np.float64)
        intr8 = np.ascontiguousarray(intr8, np.float64)
        pairs = np.zeros((frames.F, 2), np.int32)
        n, npj = C.c_int32(), C.c_int32()
        if not with_corners:
            self.ctx._ck(self.ctx.L.vsl_map_track(self.h, frames.h, int(slot), pose7.ctypes.data_as(f64p), int(model),
                                                  intr8.ctypes.data_as(f64p), int(width), int(height), C.c_double(cam_z_threshold),
                                                  C.c_double(max_dist_2d), int(threshold), C.c_double(dist_2_best),
                                                  pairs.ctypes.data_as(i32p), C.byref(n), C.byref(npj)))
            return pairs[:n.value].copy(), npj.value
        xy = np.zeros((frames.F, 2), np.float64)
        nk = C.c_int32()
        self.ctx._ck(self.ctx.L.vsl_map_track_corners(self.h, frames.h, int(slot), pose7.ctypes.data_as(f64p), int(model),
                                                      intr8.ctypes.data_as(f64p), int(width), int(height),
                                                      C.c_double(cam_z_threshold), C.c_double(max_dist_2d), int(threshold),
                                                      C.c_double(dist_2_best), pairs.ctypes.data_as(i32p), C.byref(n),
                                                      C.byref(npj), xy.ctypes.data_as(f64p), C.byref(nk)))
        return pairs[:n.value].copy(), npj.value, xy[:nk.value].copy()
