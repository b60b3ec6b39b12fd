"""CPU restatement (TEST INFRASTRUCTURE: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this) of
the per-sample work of the reference's data loader: ``HOv3Dataset._get_sample`` and what it calls
(HOIG_HOv3/data/hov3_dataset.py:16-96,113-161,215-257) -- bounding box -> affine patch transform, ``cv2.warpAffine`` of the frame
and of the resized mask, BGR->RGB / 255, ``ToTensor`` + ``Normalize`` (:267), mask / 128 (:223), ``read_obj``, ``cv2.Rodrigues``
and the posed object vertices (:239-250).

PARITY UNPINNED for the OpenCV primitives.  They live in a third-party dependency that is absent from /root/reference and from this
image -- opencv-python==4.5.1.48 (requirements.txt:99) -- and the reference holds no test or golden vector for the loader; neither can
the reference module be imported here (it imports cv2 and torchvision at module level).  What follows restates OpenCV 4.5.1's
published algorithms for exactly the calls the reference makes (modules/imgproc/src/imgwarp.cpp: getAffineTransform, warpAffine /
WarpAffineInvoker, remapBilinear with the 15-bit BilinearTab_i; resize.cpp: the 11-bit fixed-point INTER_LINEAR path for 8-bit
images; modules/calib3d/src/calibration.cpp: cvRodrigues2; modules/core/src/matrix_decomp.cpp: LUImpl behind cv::solve) and is
checked by known answers only (tests/test_data_cpu.py: identity and integer-shift warps, same-size and 2x resizes, hand-computed
pixels, rotations about the axes).  ``read_obj`` and the pure-numpy helpers are restated from the reference file itself.

Everything is numpy; integer work is exact, the floating-point steps keep the reference's dtypes (float32 transforms, float64 object
vertices rounded to float32 on assignment)."""
import numpy as np

INTER_BITS, INTER_TAB_SIZE = 5, 32
AB_BITS, AB_SCALE = 10, 1 << 10
REMAP_COEF_BITS = 15
RESIZE_COEF_BITS, RESIZE_COEF_SCALE = 11, 1 << 11


def rotate_2d(pt_2d, rot_rad):                                    # hov3_dataset.py:16-22
    x, y = pt_2d[0], pt_2d[1]
    sn, cs = np.sin(rot_rad), np.cos(rot_rad)
    return np.array([x * cs - y * sn, x * sn + y * cs], dtype=np.float32)


def _lu_solve(a, b):
    """cv::solve(A, B, X, DECOMP_LU) for one right-hand side: hal::LU64f = LUImpl<double> (partial pivoting, the elimination and the
    back substitution in OpenCV's operation order)."""
    a = [[float(v) for v in row] for row in a]
    b = [float(v) for v in b]
    m = len(b)
    for i in range(m):
        k = i
        for j in range(i + 1, m):
            if abs(a[j][i]) > abs(a[k][i]):
                k = j
        if abs(a[k][i]) < np.finfo(np.float64).eps * 100:
            raise np.linalg.LinAlgError('singular')
        if k != i:
            a[i], a[k] = a[k], a[i]
            b[i], b[k] = b[k], b[i]
        d = -1.0 / a[i][i]
        for j in range(i + 1, m):
            alpha = a[j][i] * d
            for c in range(i + 1, m):
                a[j][c] += alpha * a[i][c]
            b[j] += alpha * b[i]
    for i in range(m - 1, -1, -1):
        s = b[i]
        for c in range(i + 1, m):
            s -= a[i][c] * b[c]
        b[i] = s / a[i][i]
    return np.array(b, dtype=np.float64)


def get_affine_transform(src, dst):
    """cv2.getAffineTransform(src, dst) (three float32 point pairs) -> (2, 3) float64."""
    a = np.zeros((6, 6), np.float64)
    b = np.zeros(6, np.float64)
    for i in range(3):
        a[2 * i, 0:3] = (src[i][0], src[i][1], 1.0)
        a[2 * i + 1, 3:6] = (src[i][0], src[i][1], 1.0)
        b[2 * i], b[2 * i + 1] = dst[i][0], dst[i][1]
    return _lu_solve(a, b).reshape(2, 3)


def gen_trans_from_patch_cv(c_x, c_y, src_width, src_height, dst_width, dst_height, scale, rot, inv=False):     # :25-60
    src_w, src_h = src_width * scale, src_height * scale
    src_center = np.array([c_x, c_y], dtype=np.float32)
    rot_rad = np.pi * rot / 180
    src_downdir = rotate_2d(np.array([0, src_h * 0.5], dtype=np.float32), rot_rad)
    src_rightdir = rotate_2d(np.array([src_w * 0.5, 0], dtype=np.float32), rot_rad)
    dst_center = np.array([dst_width * 0.5, dst_height * 0.5], dtype=np.float32)
    dst_downdir = np.array([0, dst_height * 0.5], dtype=np.float32)
    dst_rightdir = np.array([dst_width * 0.5, 0], dtype=np.float32)
    src = np.zeros((3, 2), dtype=np.float32)
    src[0, :], src[1, :], src[2, :] = src_center, src_center + src_downdir, src_center + src_rightdir
    dst = np.zeros((3, 2), dtype=np.float32)
    dst[0, :], dst[1, :], dst[2, :] = dst_center, dst_center + dst_downdir, dst_center + dst_rightdir
    trans = get_affine_transform(dst, src) if inv else get_affine_transform(src, dst)
    return trans.astype(np.float32)


def patch_transform(bbox, out_shape=(256, 256)):
    """The forward 2x3 float32 transform of ``augmentation(img, bbox)`` (:87-91: scale 1, no rotation, no flip; :63-84)."""
    bb_c_x, bb_c_y = float(bbox[0] + 0.5 * bbox[2]), float(bbox[1] + 0.5 * bbox[3])
    return gen_trans_from_patch_cv(bb_c_x, bb_c_y, float(bbox[2]), float(bbox[3]), out_shape[1], out_shape[0], 1.0, 0.0)


def invert_affine(m):
    """The in-place inversion at the top of cv::warpAffine (no WARP_INVERSE_MAP), double."""
    m = np.array(m, dtype=np.float64).reshape(6)
    d = m[0] * m[4] - m[1] * m[3]
    d = 1.0 / d if d != 0 else 0.0
    a11, a22 = m[4] * d, m[0] * d
    m[0] = a11
    m[1] *= -d
    m[3] *= -d
    m[4] = a22
    b1 = -m[0] * m[2] - m[1] * m[5]
    b2 = -m[3] * m[2] - m[4] * m[5]
    m[2], m[5] = b1, b2
    return m


def _cv_round(v):
    """saturate_cast<int>(double) = cvRound: to nearest, ties to even, saturated to int32."""
    return np.clip(np.rint(v), -2147483648.0, 2147483647.0).astype(np.int64)


def warp_affine_linear_u8(img, m, dsize):
    """cv2.warpAffine(img, m, dsize, flags=cv2.INTER_LINEAR) for an 8-bit (H, W, C) image, BORDER_CONSTANT 0.  dsize = (width,
    height).  -> (height, width, C) uint8."""
    hs, ws, c = img.shape
    wd, hd = int(dsize[0]), int(dsize[1])
    mi = invert_affine(m)
    xs = np.arange(wd, dtype=np.float64)
    adelta, bdelta = _cv_round(mi[0] * xs * AB_SCALE), _cv_round(mi[3] * xs * AB_SCALE)
    ys = np.arange(hd, dtype=np.float64)
    round_delta = AB_SCALE // INTER_TAB_SIZE // 2
    x0 = _cv_round((mi[1] * ys + mi[2]) * AB_SCALE) + round_delta
    y0 = _cv_round((mi[4] * ys + mi[5]) * AB_SCALE) + round_delta
    wrap = lambda v: ((v + 2 ** 31) % 2 ** 32) - 2 ** 31                      # int arithmetic of the C code
    xf = wrap(x0[:, None] + adelta[None, :]) >> (AB_BITS - INTER_BITS)
    yf = wrap(y0[:, None] + bdelta[None, :]) >> (AB_BITS - INTER_BITS)
    sx = np.clip(xf >> INTER_BITS, -32768, 32767)                          # saturate_cast<short>
    sy = np.clip(yf >> INTER_BITS, -32768, 32767)
    fx, fy = xf & (INTER_TAB_SIZE - 1), yf & (INTER_TAB_SIZE - 1)
    # BilinearTab_i: (1-fy)(1-fx), (1-fy)fx, fy(1-fx), fy*fx at 1/32 steps, times 2^15 (exact); the entry of fx = fy = 0 saturates to
    # 32767 and initInterTab2D's sum fix-up puts the missing unit on the LAST tap
    w = np.stack([(32 - fy) * (32 - fx), (32 - fy) * fx, fy * (32 - fx), fy * fx], axis=-1).astype(np.int64) * 32
    whole = (fx == 0) & (fy == 0)
    w[whole] = (32767, 0, 0, 1)
    out = np.zeros((hd, wd, c), np.int64)
    src = img.astype(np.int64)
    for k, (dy, dx) in enumerate(((0, 0), (0, 1), (1, 0), (1, 1))):
        yy, xx = sy + dy, sx + dx
        ok = (yy >= 0) & (yy < hs) & (xx >= 0) & (xx < ws)
        tap = src[np.clip(yy, 0, hs - 1), np.clip(xx, 0, ws - 1)] * ok[..., None]
        out += tap * w[..., k][..., None]
    return ((out + (1 << (REMAP_COEF_BITS - 1))) >> REMAP_COEF_BITS).clip(0, 255).astype(np.uint8)


def resize_linear_u8(img, dsize):
    """cv2.resize(img, dsize) (INTER_LINEAR) for an 8-bit (H, W, C) image: resize.cpp's fixed-point path (11-bit coefficients,
    HResizeLinear into int rows, VResizeLinear<uchar>'s (b0*(S0>>4))>>16 form).  dsize = (width, height)."""
    hs, ws, c = img.shape
    wd, hd = int(dsize[0]), int(dsize[1])
    if (wd, hd) == (ws, hs):
        return img.copy()
    scale_x, scale_y = 1.0 / (float(wd) / ws), 1.0 / (float(hd) / hs)

    def axis(n_dst, n_src, scale, clamp_fraction):
        f = ((np.arange(n_dst, dtype=np.float64) + 0.5) * scale - 0.5).astype(np.float32)
        s = np.floor(f).astype(np.int64)
        f = (f - s.astype(np.float32)).astype(np.float32)
        if clamp_fraction:                                  # x: out-of-range taps lose their fraction
            lo, hi = s < 0, s >= n_src - 1
            f = np.where(lo | hi, np.float32(0), f)
            s = np.where(lo, 0, np.where(hi, n_src - 1, s))
        c0 = np.clip(np.rint((np.float32(1) - f) * np.float32(RESIZE_COEF_SCALE)), -32768, 32767).astype(np.int64)
        c1 = np.clip(np.rint(f * np.float32(RESIZE_COEF_SCALE)), -32768, 32767).astype(np.int64)
        return s, c0, c1

    sx, a0, a1 = axis(wd, ws, scale_x, True)
    sy, b0, b1 = axis(hd, hs, scale_y, False)
    src = img.astype(np.int64)
    x1 = np.minimum(sx + 1, ws - 1)                                        # (past xmax the row value is S[sx] * 2048: a1 is 0 there)
    rows = src[:, sx] * a0[None, :, None] + src[:, x1] * a1[None, :, None]          # (hs, wd, c) int
    clip_y = lambda v: np.where(v >= 0, np.where(v < hs, v, hs - 1), 0)
    s0, s1 = rows[clip_y(sy)], rows[clip_y(sy + 1)]
    out = (((b0[:, None, None] * (s0 >> 4)) >> 16) + ((b1[:, None, None] * (s1 >> 4)) >> 16) + 2) >> 2
    return (out & 0xFF).astype(np.uint8)                                   # (the C code casts with uchar(...), not saturate_cast)


def rodrigues(rvec):
    """cv2.Rodrigues(rvec)[0]: rotation vector (3,), (3,1) or (1,3) -> (3,3), in the input's floating-point type (computed in double)."""
    r = np.asarray(rvec)
    dt = r.dtype if r.dtype in (np.float32, np.float64) else np.float64
    r = r.astype(np.float64).reshape(3)
    theta = float(np.sqrt(r[0] * r[0] + r[1] * r[1] + r[2] * r[2]))
    if theta < np.finfo(np.float64).eps:
        return np.eye(3, dtype=dt)
    c, s = np.cos(theta), np.sin(theta)
    c1, it = 1.0 - c, 1.0 / theta if theta else 0.0
    x, y, z = r * it
    rrt = np.array([[x * x, x * y, x * z], [x * y, y * y, y * z], [x * z, y * z, z * z]])
    rx = np.array([[0, -z, y], [z, 0, -x], [-y, x, 0]])
    return (c * np.eye(3) + c1 * rrt + s * rx).astype(dt)


def read_obj_vertices(text):
    """``read_obj(filename).v`` (hov3_dataset.py:116-161): the 'v' lines' first three numbers, float64 (n, 3)."""
    v = []
    for line in text.split('\n'):
        line = line.split()
        if len(line) < 2:
            continue
        if line[0] == 'v':
            v.append([float(t) for t in line[1:4]])
    return np.array(v, dtype=np.float64).reshape(-1, 3)


MAX_OBJ_VERTS = 7866                                                      # hov3_dataset.py:246


def posed_object_vertices(v, obj_rot, obj_trans):
    """:246-248: zeros((7866, 3), float32); [:n] = v @ Rodrigues(objRot).T + objTrans (float64, rounded on assignment)."""
    out = np.zeros((MAX_OBJ_VERTS, 3), dtype=np.float32)
    now = np.matmul(v, rodrigues(obj_rot).T) + obj_trans
    out[:now.shape[0]] = now
    return out


def sample_tensors(image_bgr, mask_bgr, bbox):
    """What ``_get_sample`` + ``__getitem__`` make of a decoded frame and mask (:215-223, :208-212, :267): -> image (3, 256, 256) float32
    in [-1, 1] (RGB), mask (1, 256, 256) float32, trans (2, 3) float32."""
    mask = resize_linear_u8(mask_bgr, (640, 480))                          # :219
    trans = patch_transform(bbox)
    image = warp_affine_linear_u8(image_bgr, trans, (256, 256)).astype(np.float32)        # :78-79
    maskp = warp_affine_linear_u8(mask, trans, (256, 256)).astype(np.float32)
    image_inv = (image / 255.0)[:, :, ::-1].copy()                          # float32 / python float -> float32
    mask_inv = (maskp / 128.0)[None, :, :, -1].copy()
    t = np.ascontiguousarray(image_inv.transpose(2, 0, 1))                 # ToTensor on a float ndarray: HWC -> CHW, no scaling
    t = (t - np.float32(0.5)) / np.float32(0.5)                            # Normalize: sub_(mean).div_(std)
    return t.astype(np.float32), mask_inv.astype(np.float32), trans


# ---- the HOIG_DexYCB copy (HOIG_DexYCB/data/ycb_dataset.py:132-174,281-305): no mask, corner-form boxes, the object posed by a 3x4 matrix
YCB_MAX_OBJ_VERTS = 8000


def ycb_sample_tensors(image_bgr, bbox_xyxy):
    """:284-290, :274: -> image (3, 256, 256) float32 in [-1, 1] (RGB), trans (2, 3) float32."""
    bbox = [bbox_xyxy[0], bbox_xyxy[1], bbox_xyxy[2] - bbox_xyxy[0], bbox_xyxy[3] - bbox_xyxy[1]]
    trans = patch_transform(bbox)
    image = warp_affine_linear_u8(image_bgr, trans, (256, 256)).astype(np.float32)
    image_inv = (image / 255.0)[:, :, ::-1].copy()
    t = np.ascontiguousarray(image_inv.transpose(2, 0, 1))
    t = (t - np.float32(0.5)) / np.float32(0.5)
    return t.astype(np.float32), trans


def ycb_object_vertices(v, pose_y, grasp_id):
    """:153-169, :292-293: the non-zero 3x4 poses of the label file get a fourth row, the grasped one (indexed AMONG THE NON-ZERO ones, as
    the reference does) multiplies the homogeneous vertices in float64; rounded to float32 into 8000 rows."""
    pose_obj_list = [np.vstack((pose_y[o], np.array([[0, 0, 0, 1]], dtype=np.float32))) for o in range(len(pose_y))
                     if not np.all(pose_y[o] == 0.0)]
    homo = np.concatenate([v, np.ones_like(v)[:, 2:]], axis=1)
    now = np.matmul(pose_obj_list[grasp_id], homo.T)[:3].transpose(1, 0)
    out = np.zeros((YCB_MAX_OBJ_VERTS, 3), dtype=np.float32)
    out[:now.shape[0]] = now
    return out
