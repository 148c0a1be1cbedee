"""Oracle (test infrastructure): numpy restatement of the soft-argmax decoder.

Follows the reference op for op:
  * coordinate grid ............ /root/reference/utils.py:24-35  (generate_com_filter)
  * heatmap normalisation ...... /root/reference/model.py:81-90  (softmax(w*z) | relu-sum)
  * u, v expectation ........... /root/reference/model.py:92-95
  * masked depth expectation ... /root/reference/model.py:123-129
The backward pass is the closed form of what autograd derives for those lines
(SURVEY.md section 8 a-D); ``tests/test_oracle_golden.py`` pins both directions against
vectors produced by the reference's own autograd.

Never imported by the product package.
"""
import numpy as np

EPS = 1e-14


def com_grid(P, dtype=np.float32):
    """[2,P,P] grid: ch0 = (col - P//2)/(P-1), ch1 = (row - P//2)/(P-1)  (utils.py:28-34).

    The reference builds it in float64 and casts to float32 (model.py:68); computing
    (i - P//2)/(P-1) directly in float32 gives the identical bits for the sizes in use,
    which test_oracle_golden checks.
    """
    ax = (np.arange(P, dtype=np.float64) - (P // 2)) / (P - 1)
    gx = np.broadcast_to(ax[None, :], (P, P))
    gy = np.broadcast_to(ax[:, None], (P, P))
    return np.stack([gx, gy]).astype(np.float32).astype(dtype)


def decode_forward(z, D, L, m, w=None, method="softmax", dtype=None):
    """z, D: [B,J,P,P]; L, m: [B,1,P,P]; w: [J,1] (softmax only).

    Returns heatmaps p [B,J,P,P] and uvd [B,J,3].
    """
    dtype = dtype or z.dtype
    z = z.astype(dtype); D = D.astype(dtype); L = L.astype(dtype); m = m.astype(dtype)
    B, J, P, _ = z.shape
    g = com_grid(P, dtype)
    if method == "softmax":
        x = (w.astype(dtype).reshape(1, J, 1) * z.reshape(B, J, -1))        # model.py:84
        x = x - x.max(axis=2, keepdims=True)
        e = np.exp(x)
        p = (e / e.sum(axis=2, keepdims=True)).reshape(B, J, P, P)
    else:
        r = (np.maximum(z, 0) + EPS).astype(dtype)                          # model.py:88-89
        p = r / r.sum(axis=(2, 3), keepdims=True)                           # model.py:88-90
    u = (g[0][None, None] * p).sum(axis=(2, 3))                             # model.py:92
    v = (g[1][None, None] * p).sum(axis=(2, 3))                             # model.py:93
    recon = D + L                                                           # model.py:123
    mrecon = m * recon                                                      # model.py:124
    mp = p * m                                                              # model.py:125
    d = (mp * mrecon).sum(axis=(2, 3)) / (mp.sum(axis=(2, 3)) + EPS)        # model.py:127-129
    uvd = np.stack([u, v, d], axis=2).astype(dtype)
    return p.astype(dtype), uvd


def decode_backward(z, D, L, m, w, gH, gD, gU, method="softmax", dtype=None):
    """Closed-form backward.  gH, gD: [B,J,P,P] (may be None = zeros); gU: [B,J,3].

    Returns g_z, g_D (both [B,J,P,P]) and g_w ([J,1], or None for method 'sum').
    """
    dtype = dtype or z.dtype
    z = z.astype(dtype); D = D.astype(dtype); L = L.astype(dtype); m = m.astype(dtype)
    B, J, P, _ = z.shape
    gH = np.zeros_like(z) if gH is None else gH.astype(dtype)
    gD = np.zeros_like(z) if gD is None else gD.astype(dtype)
    gU = gU.astype(dtype)
    g = com_grid(P, dtype)
    p, uvd = decode_forward(z, D, L, m, w, method, dtype)
    d = uvd[:, :, 2][:, :, None, None]
    S = (p * m).sum(axis=(2, 3), keepdims=True) + EPS
    gu = gU[:, :, 0][:, :, None, None]
    gv = gU[:, :, 1][:, :, None, None]
    gd = gU[:, :, 2][:, :, None, None]
    g_p = gH + gu * g[0][None, None] + gv * g[1][None, None] + gd * m * (m * (D + L) - d) / S
    g_D = gD + gd * p * m * m / S
    if method == "softmax":
        wv = w.astype(dtype).reshape(1, J, 1, 1)
        dot = (p * g_p).sum(axis=(2, 3), keepdims=True)
        g_y = p * (g_p - dot)
        g_z = wv * g_y
        g_w = (g_y * z).sum(axis=(0, 2, 3)).reshape(J, 1)
        return g_z.astype(dtype), g_D.astype(dtype), g_w.astype(dtype)
    r = np.maximum(z, 0) + EPS
    T = r.sum(axis=(2, 3), keepdims=True)
    dot = (p * g_p).sum(axis=(2, 3), keepdims=True)
    g_r = (g_p - dot) / T
    g_z = g_r * (z > 0)
    return g_z.astype(dtype), g_D.astype(dtype), None
