"""Host-side geometry of the loader (a few dozen flops per sample): the patch transform of ``augmentation`` (HOIG_HOv3/data/
hov3_dataset.py:25-91: scale 1, no rotation, no flip -> cv2.getAffineTransform of three point pairs), ``cv2.Rodrigues`` (:247) and
the OBJ vertex reader (:116-161)."""
import numpy as np


def rotate_2d(pt_2d, rot_rad):                                                      # hov3_dataset.py:16-22
    x, y = pt_2d[0], pt_2d[1]
    sn, cs = np.sin(rot_rad), np.cos(rot_rad)
    return np.array([x * cs - y * sn, x * sn + y * cs], dtype=np.float32)


def get_affine_transform(src, dst):
    """cv2.getAffineTransform: the 6x6 system [x y 1 0 0 0; 0 0 0 x y 1] m = [u; v] through cv::solve's default DECOMP_LU, i.e.
    OpenCV's own elimination order in double (modules/core/src/matrix_decomp.cpp LUImpl) -- a LAPACK solve differs from it in the last
    bits, which survive the float32 cast in the entries that are zero up to rounding."""
    m = 6
    a = [[0.0] * m for _ in range(m)]
    b = [0.0] * m
    for i in range(3):
        x, y = float(src[i][0]), float(src[i][1])
        a[2 * i][0], a[2 * i][1], a[2 * i][2] = x, y, 1.0
        a[2 * i + 1][3], a[2 * i + 1][4], a[2 * i + 1][5] = x, y, 1.0
        b[2 * i], b[2 * i + 1] = float(dst[i][0]), float(dst[i][1])
    for i in range(m):
        k = max(range(i, m), key=lambda j: (abs(a[j][i]), -j))          # first row of the largest pivot
        if abs(a[k][i]) < 2.220446049250313e-14:
            raise np.linalg.LinAlgError('getAffineTransform: collinear points')
        if k != i:
            a[i], a[k] = a[k], a[i]
            b[i], b[k] = b[k], b[i]
        d = -1.0 / a[i][i]
        for j in range(i + 1, m):
            alpha = a[j][i] * d
            rj, ri = a[j], a[i]
            for c in range(i + 1, m):
                rj[c] += alpha * ri[c]
            b[j] += alpha * b[i]
    for i in range(m - 1, -1, -1):
        s = b[i]
        for c in range(i + 1, m):
            s -= a[i][c] * b[c]
        b[i] = s / a[i][i]
    return np.array(b, dtype=np.float64).reshape(2, 3)


def gen_trans_from_patch_cv(c_x, c_y, src_width, src_height, dst_width, dst_height, scale, rot, inv=False):     # :25-60
    src_w, src_h = src_width * scale, src_height * scale
    src_center = np.array([c_x, c_y], dtype=np.float32)
    rot_rad = np.pi * rot / 180
    src = np.zeros((3, 2), dtype=np.float32)
    src[0] = src_center
    src[1] = src_center + rotate_2d(np.array([0, src_h * 0.5], dtype=np.float32), rot_rad)
    src[2] = src_center + rotate_2d(np.array([src_w * 0.5, 0], dtype=np.float32), rot_rad)
    dst = np.zeros((3, 2), dtype=np.float32)
    dst[0] = np.array([dst_width * 0.5, dst_height * 0.5], dtype=np.float32)
    dst[1] = dst[0] + np.array([0, dst_height * 0.5], dtype=np.float32)
    dst[2] = dst[0] + np.array([dst_width * 0.5, 0], dtype=np.float32)
    m = get_affine_transform(dst, src) if inv else get_affine_transform(src, dst)
    return m.astype(np.float32)


def patch_transform(bbox, out_shape=(256, 256)):
    """``augmentation(img, bbox)[1]`` (:87-91,63-84): the 2x3 float32 transform frame -> 256 x 256 patch."""
    bb_c_x, bb_c_y = float(bbox[0] + 0.5 * bbox[2]), float(bbox[1] + 0.5 * bbox[3])
    return gen_trans_from_patch_cv(bb_c_x, bb_c_y, float(bbox[2]), float(bbox[3]), out_shape[1], out_shape[0], 1.0, 0.0)


def rodrigues(rvec):
    """cv2.Rodrigues(rvec)[0] (rotation vector -> matrix), computed in double, returned in the input's floating type."""
    r = np.asarray(rvec)
    dt = r.dtype if r.dtype in (np.float32, np.float64) else np.float64
    r = r.astype(np.float64).reshape(3)
    theta = float(np.sqrt(r[0] * r[0] + r[1] * r[1] + r[2] * r[2]))
    if theta < np.finfo(np.float64).eps:
        return np.eye(3, dtype=dt)
    c, s = np.cos(theta), np.sin(theta)
    x, y, z = r * (1.0 / theta)
    rrt = np.array([[x * x, x * y, x * z], [x * y, y * y, y * z], [x * z, y * z, z * z]])
    rx = np.array([[0, -z, y], [z, 0, -x], [-y, x, 0]])
    return (c * np.eye(3) + (1.0 - c) * rrt + s * rx).astype(dt)


def read_obj_vertices(filename):
    """``read_obj(filename).v`` (:116-161): the first three numbers of every 'v' line, float64 (n, 3).  Parsed ONCE per mesh here
    (MeshCache); the reference re-reads and re-parses the file for every item (:239)."""
    rows = []
    with open(filename) as f:
        for line in f:
            t = line.split()
            if len(t) >= 2 and t[0] == 'v':
                rows.append((float(t[1]), float(t[2]), float(t[3])))
    return np.array(rows, dtype=np.float64).reshape(-1, 3)
