"""CPU restatement of the reference's input pipeline (TEST INFRASTRUCTURE ONLY, see oracle/__init__.py):
/root/reference/datasets.py:200-403 (HandDataset.process_single_data, mode 'uvd') and the helpers it calls
(utils.center_crop utils.py:167-173, utils.random_rotated utils.py:66-82).

The three OpenCV calls on that path are restated from OpenCV's published algorithms (cv2 is not installed in this image and the
reference's env.yml does not pin it):
  cv2.resize(src, (w, h))                      INTER_LINEAR, float32: horizontal pass then vertical pass, pixel centres aligned
                                               (fx = (dx + 0.5) * scale - 0.5), source index clamped to the image       -> resize_linear
  cv2.getRotationMatrix2D(center, angle, scale)                                                                           -> rotation_matrix
  cv2.warpAffine(src, M, (w, h))               INTER_LINEAR, BORDER_CONSTANT(0): M is inverted, source coordinates in fixed point
                                               (AB_BITS = 10, rounded to 1/32 pixel), bilinear weights from the 32x32 table -> warp_affine
Parity of these three is therefore pinned to the published algorithm; everything around them (crop, depth threshold, centring
on the COM, joint transforms, normalisation, the un-augmented fallback) is pinned to the reference itself: oracle/gen_golden.py
runs HandDataset.process_single_data with `cv2` stubbed by exactly these functions (tests/golden/preprocess.npz).
"""
import numpy as np


def resize_linear(src, dsize):
    """cv2.resize(src, dsize=(width, height)) for a 2-D float32 image, INTER_LINEAR."""
    src = np.asarray(src, dtype=np.float32)
    H, W = src.shape
    dw, dh = int(dsize[0]), int(dsize[1])
    sx, sy = W / dw, H / dh                                        # double, like inv_scale_x / inv_scale_y

    def taps(n_dst, n_src, scale):
        idx = np.zeros(n_dst, np.int64); a1 = np.zeros(n_dst, np.float32)
        for d in range(n_dst):
            f = np.float32((d + 0.5) * scale - 0.5)
            s = int(np.floor(f)); f = np.float32(f - s)
            if s < 0:
                s, f = 0, np.float32(0)
            if s >= n_src - 1:
                s, f = n_src - 1, np.float32(0)
            idx[d], a1[d] = s, f
        return idx, a1
    xi, xa = taps(dw, W, sx)
    yi, ya = taps(dh, H, sy)
    xi1 = np.minimum(xi + 1, W - 1)
    rows = src[:, xi] * (np.float32(1) - xa)[None, :] + src[:, xi1] * xa[None, :]      # HResize (float32)
    yi1 = np.minimum(yi + 1, H - 1)
    return (rows[yi] * (np.float32(1) - ya)[:, None] + rows[yi1] * ya[:, None]).astype(np.float32)   # VResize


def rotation_matrix(center, angle, scale):
    """cv2.getRotationMatrix2D: [[a, b, (1-a)cx - b cy], [-b, a, b cx + (1-a) cy]], a = scale cos, b = scale sin (degrees)."""
    ang = angle * np.pi / 180.0
    a, b = scale * np.cos(ang), scale * np.sin(ang)
    cx, cy = float(center[0]), float(center[1])
    return np.array([[a, b, (1 - a) * cx - b * cy], [-b, a, b * cx + (1 - a) * cy]], dtype=np.float64)


def invert_affine(M):
    """What cv2.warpAffine does to M (no WARP_INVERSE_MAP): dst -> src map."""
    M = np.asarray(M, dtype=np.float64)
    D = M[0, 0] * M[1, 1] - M[0, 1] * M[1, 0]
    D = 1.0 / D if D != 0 else 0.0
    A11, A22 = M[1, 1] * D, M[0, 0] * D
    i00, i01, i10, i11 = A11, -M[0, 1] * D, -M[1, 0] * D, A22
    b1 = -i00 * M[0, 2] - i01 * M[1, 2]
    b2 = -i10 * M[0, 2] - i11 * M[1, 2]
    return np.array([[i00, i01, b1], [i10, i11, b2]], dtype=np.float64)


AB_BITS, INTER_BITS = 10, 5
AB_SCALE, INTER_TAB_SIZE = 1 << AB_BITS, 1 << INTER_BITS


def _cv_round(x):
    return np.rint(x).astype(np.int64)          # cvRound: round half to even


def warp_affine(src, M, dsize):
    """cv2.warpAffine(src, M, dsize=(width, height)), INTER_LINEAR, BORDER_CONSTANT with value 0, float32 image."""
    src = np.asarray(src, dtype=np.float32)
    H, W = src.shape
    dw, dh = int(dsize[0]), int(dsize[1])
    Mi = invert_affine(M)
    xs = np.arange(dw)
    adelta = _cv_round(Mi[0, 0] * xs * AB_SCALE)
    bdelta = _cv_round(Mi[1, 0] * xs * AB_SCALE)
    round_delta = AB_SCALE // INTER_TAB_SIZE // 2
    out = np.zeros((dh, dw), np.float32)
    pad = np.zeros((H + 2, W + 2), np.float32)
    pad[1:-1, 1:-1] = src
    for y in range(dh):
        X0 = int(_cv_round((Mi[0, 1] * y + Mi[0, 2]) * AB_SCALE)) + round_delta
        Y0 = int(_cv_round((Mi[1, 1] * y + Mi[1, 2]) * AB_SCALE)) + round_delta
        X = (X0 + adelta) >> (AB_BITS - INTER_BITS)
        Y = (Y0 + bdelta) >> (AB_BITS - INTER_BITS)
        ix, iy = X >> INTER_BITS, Y >> INTER_BITS
        fx = ((X & (INTER_TAB_SIZE - 1)) / np.float32(INTER_TAB_SIZE)).astype(np.float32)
        fy = ((Y & (INTER_TAB_SIZE - 1)) / np.float32(INTER_TAB_SIZE)).astype(np.float32)
        ok = (ix >= -1) & (ix < W) & (iy >= -1) & (iy < H)
        cx, cy = np.clip(ix, -1, W - 1) + 1, np.clip(iy, -1, H - 1) + 1
        s00, s01, s10, s11 = pad[cy, cx], pad[cy, cx + 1], pad[cy + 1, cx], pad[cy + 1, cx + 1]
        w00 = (np.float32(1) - fy) * (np.float32(1) - fx); w01 = (np.float32(1) - fy) * fx
        w10 = fy * (np.float32(1) - fx); w11 = fy * fx
        out[y] = np.where(ok, s00 * w00 + s01 * w01 + s10 * w10 + s11 * w11, np.float32(0))
    return out


def center_crop(img, center, window):
    """utils.py:167-173 (note: `center` is (row, col); the window is 2 * (window // 2) wide)."""
    u, v = int(center[0]), int(center[1])
    shift = window // 2
    d = np.pad(img, ((shift, shift), (shift, shift)), "constant", constant_values=0)
    return d[u:u + 2 * shift, v:v + 2 * shift]


def process_single(image, joint_uvd, com, cube_size, fx, fy, image_size=128, label_size=64, angle=None, scale=1.0):
    """datasets.py:243-403 for one sample.  angle is None -> the un-augmented path (datasets.py:300-362); otherwise rotation by
    `angle` degrees and `scale` as utils.random_rotated applies them (the angle that function draws itself, utils.py:70).
    Returns dict: img, label_img, mask (un-normalised label != 0), box_size, com (u, v truncated), uvd (normalised)."""
    image = np.asarray(image, dtype=np.float32); com = np.array(com, dtype=np.float64)
    du, dv = cube_size / com[2] * fx, cube_size / com[2] * fy
    box = max(int(du + dv), 2)
    crop = center_crop(image, (com[1], com[0]), box).astype(np.float32)
    crop = crop * np.logical_and(crop > com[2] - cube_size, crop < com[2] + cube_size)
    crop = crop.astype(np.float32)
    crop[crop > 0] -= np.float32(com[2])
    com[0], com[1] = int(com[0]), int(com[1])
    box = crop.shape[0]
    img = resize_linear(crop, (image_size, image_size))
    cen = np.asarray(joint_uvd, dtype=np.float64) - com
    cen[:, :2] = cen[:, :2] / (box - 1) * (image_size - 1)
    if angle is not None:
        M = rotation_matrix((image_size // 2, image_size // 2), angle, scale)
        img = warp_affine(img, M, (image_size, image_size))
        a = angle / 180.0 * np.pi
        Rot = np.array([[np.cos(a), np.sin(a)], [-np.sin(a), np.cos(a)]])
        cen[:, :2] = cen[:, :2] @ Rot.T
        cen[:, :2] = cen[:, :2] * scale
        img = (img * np.float32(scale)).astype(np.float32) if not isinstance(scale, float) else (img * scale)
        cen[:, 2] *= scale
    label = resize_linear(np.asarray(img, dtype=np.float32), (label_size, label_size))
    mask = (label != 0).astype(np.float64)
    uvd = cen.copy()
    uvd[:, :2] /= (image_size - 1)
    uvd[:, 2] /= cube_size
    return {"img": np.asarray(img) / cube_size, "label_img": label / cube_size, "mask": mask, "box_size": box, "com": com, "uvd": uvd,
            "label_raw": label, "joint_centered_resized": cen}


def sample_like_reference(image, joint_uvd, com, cube_size, fx, fy, image_size=128, label_size=64, aug=None):
    """The control flow of HandDataset.process_single_data around process_single (datasets.py:221-390):
      * with augmentation: the augmented attempt sits in a try (datasets.py:221-299); the only thing that fails in it for valid
        frames is utils.generate_heatmap (joint footprint out of numpy's index range, or a NaN position) -> the bare `except`
        (:300) re-does the sample WITHOUT augmentation ("fallback");
      * the un-augmented path raises ValueError when generate_heatmap fails there (:358-365);
      * whichever path produced the sample, it raises ValueError when anything is NaN or sum(mask) < 10 (:385-390).
    A ValueError means the dataset drops the sample (check_text, datasets.py:159-167): "rejected".
    aug: dict(angle, scale, shift_x, shift_y) or None.  Returns (process_single's dict or None, fallback, rejected)."""
    from oracle import targets_ref as TR

    def heat_ok(o):
        uv = TR.label_pixels(o["uvd"], label_size)
        return all(TR.footprint_ok(label_size, uv[j, 0], uv[j, 1]) for j in range(uv.shape[0]))
    o, fallback = None, False
    if aug is not None:
        o = process_single(image, joint_uvd, shift_com(com, aug["shift_x"], aug["shift_y"]), cube_size, fx, fy, image_size, label_size,
                           angle=aug["angle"], scale=aug["scale"])
        if not heat_ok(o):
            o, fallback = None, True
    if o is None:
        o = process_single(image, joint_uvd, com, cube_size, fx, fy, image_size, label_size)
        if not heat_ok(o):
            return None, fallback, True
    bad = any(np.isnan(np.asarray(o[k], dtype=np.float64)).any() for k in ("img", "label_img", "uvd", "mask")) or o["mask"].sum() < 10
    return (None if bad else o), fallback, bool(bad)


def shift_com(com, shift_x, shift_y):
    """datasets.py:235-241.  The reference means to shift the COM in camera space (uvd2xyz -> += shift -> xyz2uvd), but
    HandDataset.uvd2xyz / xyz2uvd (datasets.py:85-111) only transform 2-D and 3-D arrays and return a 1-D [u, v, d] vector
    unchanged -- so what it really does is shift the COM by (shift_x, shift_y) PIXELS.  Restated as it behaves."""
    c = np.array(com, dtype=np.float64)
    c[0] += shift_x
    c[1] += shift_y
    return c


def draws_to_augmentation(draws):
    """The reference's python-`random` draws in order (datasets.py:225-238, utils.py:70): [angle (drawn, never used), scale, shift_x,
    shift_y, angle actually applied by utils.random_rotated]."""
    return {"scale": 0.8 + draws[1] * 0.4, "shift_x": -5 + draws[2] * 10, "shift_y": -5 + draws[3] * 10, "angle": draws[4] * 60 - 30}
