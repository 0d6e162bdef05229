"""Independent numpy restatement of crates/chalkydri_sqpnp/src/lib.rs (numpy.linalg for SVD / eigh / solve) and
scene helpers.  Used only to pin oracle/sqpnp.c and the HIP solver; citations are to the reference file."""
import numpy as np

TAG_SIZE = 0.1651
S = TAG_SIZE / 2.0
CORNERS = np.array([[0, -S, -S], [0, S, -S], [0, S, S], [0, -S, S]], float)  # lib.rs:383-388


def quat_to_mat(q):
    w, x, y, z = np.asarray(q, float) / np.linalg.norm(q)
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def mat_to_quat(R):
    from scipy.spatial.transform import Rotation
    x, y, z, w = Rotation.from_matrix(R).as_quat()
    return np.array([w, x, y, z])


def euler_to_mat(roll, pitch, yaw):
    cr, sr, cp, sp, cy, sy = np.cos(roll), np.sin(roll), np.cos(pitch), np.sin(pitch), np.cos(yaw), np.sin(yaw)
    Rz = np.array([[cy, -sy, 0], [sy, cy, 0], [0, 0, 1]])
    Ry = np.array([[cp, 0, sp], [0, 1, 0], [-sp, 0, cp]])
    Rx = np.array([[1, 0, 0], [0, cr, -sr], [0, sr, cr]])
    return Rz @ Ry @ Rx


NWU_TO_CV = np.array([[0, 0, 1], [-1, 0, 0], [0, -1, 0]], float)  # lib.rs:449-453


def create_solver_camera_transform(fwd, left, up, roll_deg, pitch_deg, yaw_deg):
    """lib.rs:430-461 -> (R, t) of cam_cv <- robot."""
    Rn = euler_to_mat(*np.radians([roll_deg, pitch_deg, yaw_deg]))
    Rc = Rn @ NWU_TO_CV
    T = np.array([fwd, left, up], float)
    return Rc.T, -Rc.T @ T


def _random_orthogonal(rng, k):
    q, rr = np.linalg.qr(rng.normal(size=(k, k)))
    return q * np.sign(np.diag(rr))


def nearest_so3(r, rng_basis=None):  # lib.rs:42-59 (r column-major)
    M = r.reshape(3, 3).T
    U, sv, Vt = np.linalg.svd(M)
    if rng_basis is not None:
        # a singular value of (numerically) zero leaves its columns of U and V free: any SVD routine may return any
        # orthonormal completion.  Draw one at random, independently for U and V.
        null = sv <= 1e-12 * max(sv[0], 1e-300)
        k = int(null.sum())
        if k >= 1:
            U = U.copy(); Vt = Vt.copy()
            U[:, null] = U[:, null] @ _random_orthogonal(rng_basis, k)
            Vt[null, :] = _random_orthogonal(rng_basis, k) @ Vt[null, :]
    R = U @ Vt
    if np.linalg.det(R) < 0:
        U[:, 2] *= -1
        R = U @ Vt
    return R.T.reshape(9)


def constraints(r):  # lib.rs:62-95
    c1, c2, c3 = r[0:3], r[3:6], r[6:9]
    h = np.array([c1 @ c1 - 1, c2 @ c2 - 1, c3 @ c3 - 1, c1 @ c2, c1 @ c3, c2 @ c3])
    J = np.zeros((6, 9))
    J[0, 0:3] = 2 * c1; J[1, 3:6] = 2 * c2; J[2, 6:9] = 2 * c3
    J[3, 0:3] = c2; J[3, 3:6] = c1; J[4, 0:3] = c3; J[4, 6:9] = c1; J[5, 3:6] = c3; J[5, 6:9] = c2
    return h, J


def optimization(r, omega, max_iter=15, tol_sq=1e-16):  # lib.rs:463-480, 98-115
    r = r.copy()
    for _ in range(max_iter):
        h, J = constraints(r)
        lhs = np.zeros((15, 15))
        lhs[:9, :9] = omega; lhs[:9, 9:] = J.T; lhs[9:, :9] = J
        rhs = np.concatenate([-omega @ r, -h])
        try:
            sol = np.linalg.solve(lhs, rhs)
        except np.linalg.LinAlgError:
            break
        d = sol[:9]
        r += d
        if d @ d < tol_sq:
            break
    return r, r @ omega @ r


def build_linear_system(p3, p2):  # lib.rs:124-180
    q_rr = np.zeros((9, 9)); q_rt = np.zeros((9, 3)); q_tt = np.zeros((3, 3))
    for X, v in zip(p3, p2):
        P = np.eye(3) - np.outer(v, v) / (v @ v)
        q_tt += P
        for a in range(3):
            q_rt[3 * a:3 * a + 3] += P * X[a]
            for b in range(3):
                q_rr[3 * a:3 * a + 3, 3 * b:3 * b + 3] += P * X[a] * X[b]
    q_tt_inv = np.linalg.inv(q_tt) if abs(np.linalg.det(q_tt)) > 0 else np.zeros((3, 3))
    omega = q_rr - q_rt @ q_tt_inv @ q_rt.T
    return omega, q_tt_inv, q_rt


def solve_robot_pose(tags, bearings, robot_to_cam, gyro, sign_change_error=600.0, max_iter=15, tol_sq=1e-16, rng_basis=None):
    """tags: list of (R 3x3, t 3); bearings: (4n,3); robot_to_cam: (R,t).  Returns dict or None (lib.rs:297-377).
    rng_basis: when given, every choice a linear-algebra library is free to make differently is drawn at random — the
    orthonormal basis of each (numerically) degenerate eigenspace of Omega, the order of equal eigenvalues, the signs of
    the eigenvectors, and the completion of rank-deficient SVDs in nearest_so3 — to test that the returned pose does not
    depend on them (the reference uses nalgebra's symmetric_eigen / svd, chalkydri_sqpnp/src/lib.rs:45,398)."""
    Rrc, trc = robot_to_cam
    fwd = Rrc[:, 0]
    world = np.concatenate([(R @ CORNERS.T).T + t for R, t in tags])
    n = len(world)
    if n < 3 or n != len(bearings):
        return None
    centroid = world.mean(0)
    omega, q_tt_inv, q_rt = build_linear_system(world - centroid, bearings)
    w, V = np.linalg.eigh(omega)
    order = np.argsort(w, kind="stable")
    if rng_basis is not None:
        V = V * rng_basis.choice([-1.0, 1.0], size=9)
        scale = max(abs(w).max(), 1e-300)
        ws = w[order]
        start = 0
        for end in range(1, 10):                      # clusters of eigenvalues closer than 1e-9 of the spectrum's scale
            if end == 9 or ws[end] - ws[end - 1] > 1e-9 * scale:
                if end - start > 1:
                    idx = order[start:end]
                    V[:, idx] = V[:, idx] @ _random_orthogonal(rng_basis, end - start)
                    order[start:end] = rng_basis.permutation(idx)
                start = end
    cands = []
    for i in order[:3]:
        for sign in (-1.0, 1.0):
            r, e = optimization(nearest_so3(V[:, i] * sign, rng_basis), omega, max_iter, tol_sq)
            fx, fy = r[0:3] @ fwd, r[3:6] @ fwd
            e += sign_change_error * max(0.0, 1.0 - (fx * np.cos(gyro) + fy * np.sin(gyro)))
            cands.append((r, e))
    cands.sort(key=lambda c: c[1])
    best = None
    for r, e in cands:
        Rm = r.reshape(3, 3).T
        t = -(q_tt_inv @ (q_rt.T @ r)) - Rm @ centroid
        if np.all((world @ Rm.T + t)[:, 2] > 0):
            U, _, Vt = np.linalg.svd(Rm)
            Rn = U @ Vt
            if np.linalg.det(Rn) < 0:
                U[:, 2] *= -1
                Rn = U @ Vt
            best = (Rn, t, r @ omega @ r)
            break
    if best is None:
        return None
    Rwc, twc, energy = best
    n_tags = len(tags)
    with np.errstate(invalid="ignore"):
        rms = np.sqrt(energy / (4 * n_tags))  # NaN for a (round-off) negative energy, exactly like f64::sqrt
    dist = np.linalg.norm(twc)
    if rms > 0.1:
        std = np.full(3, np.finfo(float).max)
    elif np.isnan(rms):
        std = np.full(3, np.nan)  # f64::clamp keeps NaN
    else:
        m = 1 + dist / TAG_SIZE
        xy = np.clip(rms * m / np.sqrt(n_tags) * 5.0, 0.01, 10.0)
        th = np.clip(rms / TAG_SIZE * m / np.sqrt(n_tags) * 2.0, 0.05, np.pi)
        std = np.array([xy, xy, th])
    robot_rot = Rwc.T @ Rrc
    robot_pos = Rwc.T @ (trc - twc)
    tc = np.mean([t for _, t in tags], 0)
    vyaw = np.arctan2(robot_rot[1, 0], robot_rot[0, 0])
    d = (gyro - vyaw + np.pi) % (2 * np.pi) - np.pi
    wgt = np.clip(abs(np.degrees(d)) / 30.0, 0, 1)
    wgt = wgt * wgt * (3 - 2 * wgt)
    a = d * wgt
    Rz = np.array([[np.cos(a), -np.sin(a), 0], [np.sin(a), np.cos(a), 0], [0, 0, 1]])
    rot = Rz @ robot_rot
    pos = tc + Rz @ (robot_pos - tc)
    yaw = np.arctan2(rot[1, 0], rot[0, 0]) if abs(rot[2, 0]) < 1 else 0.0
    return {"rot": rot, "pos": pos, "std": std, "yaw": yaw, "energy": energy}


def make_scene(rng, n_tags, noise_px=0.0, f=1000.0):
    """Random robot pose looking at a wall of tags.  Returns tags [(R,t)], bearings (4n,3), robot_to_cam (R,t), truth."""
    robot_to_cam = create_solver_camera_transform(rng.uniform(-0.3, 0.3), rng.uniform(-0.3, 0.3), rng.uniform(0.2, 0.8),
                                                  rng.uniform(-5, 5), rng.uniform(-15, 15), rng.uniform(-30, 30))
    yaw = rng.uniform(-np.pi, np.pi)
    Rwr = euler_to_mat(0, 0, yaw)
    twr = np.array([rng.uniform(2, 14), rng.uniform(1, 7), 0.0])
    Rrc, trc = robot_to_cam                      # cam <- robot
    Rcw = Rrc @ Rwr.T                            # cam <- world
    tcw = trc - Rcw @ twr
    tags = []
    for _ in range(n_tags):
        # tag in front of the camera, facing it: tag-local x is the tag normal (lib.rs:383-388)
        pc = np.array([rng.uniform(-1.5, 1.5), rng.uniform(-0.8, 0.8), rng.uniform(2.0, 6.0)])
        Rtc = euler_to_mat(rng.uniform(-0.5, 0.5), rng.uniform(-0.5, 0.5), rng.uniform(-0.5, 0.5)) @ np.array([[0, 0, 1], [-1, 0, 0], [0, -1, 0]], float).T
        Rtw = Rcw.T @ Rtc
        ttw = Rcw.T @ (pc - tcw)
        tags.append((Rtw, ttw))
    world = np.concatenate([(R @ CORNERS.T).T + t for R, t in tags])
    cam = world @ Rcw.T + tcw
    px = cam[:, :2] / cam[:, 2:3] * f
    px += rng.normal(0, noise_px, px.shape) if noise_px > 0 else 0
    b = np.concatenate([px / f, np.ones((len(px), 1))], 1)
    b /= np.linalg.norm(b, axis=1, keepdims=True)
    return tags, b, robot_to_cam, {"Rwr": Rwr, "twr": twr, "yaw": yaw}
