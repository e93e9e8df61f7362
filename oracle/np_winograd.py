"""ORACLE (test infrastructure, not product code) — the Winograd F(4x4,3x3) / F(3x3,4x4) algebra
of csrc/conv_winograd.h, restated in NumPy.

What it pins: the transform constants the HIP kernels carry (interpolation points 0, 1, -1, 1/2,
-2, infinity).  The matrices are DERIVED here from the points by exact rational arithmetic
(Cook-Toom / Lagrange construction: A^T is the Vandermonde matrix of the points, G its scaled
counterpart on the filter side, B^T follows from the bilinear identity), and the test
(tests/test_oracle_winograd.py) checks (a) that the derived matrices equal the literal ones the
kernels use, (b) that the three passes reproduce the direct convolution of oracle/np_ref.py
(conv2d_fwd / conv2d_bwd — the restatement of chainer's L.Convolution2D, reference call sites
/root/reference/chainer_mask_rcnn/models/mask_rcnn_resnet.py:131-143 and
models/region_proposal_network.py:75-80) to float64 round-off on ragged shapes.
"""
from fractions import Fraction as Fr

import numpy as np

POINTS = (Fr(0), Fr(1), Fr(-1), Fr(1, 2), Fr(-2))     # + the point at infinity

# literal copies of what the kernels use (wino_bt / wino_at / kFilterG / wino_g4 / kWgradAT)
BT = np.array([[1, -1.5, -2, 1.5, 1, 0],
               [0, -1, 0.5, 2.5, 1, 0],
               [0, 1, -2.5, 0.5, 1, 0],
               [0, -2, -1, 2, 1, 0],
               [0, 0.5, -1, -0.5, 1, 0],
               [0, 1, -1.5, -2, 1.5, 1]], np.float64)
AT = np.array([[1, 1, 1, 1, 1, 0],
               [0, 1, -1, 0.5, -2, 0],
               [0, 1, 1, 0.25, 4, 0],
               [0, 1, -1, 0.125, -8, 1]], np.float64)
G = np.array([[1, 0, 0],
              [1 / 3, 1 / 3, 1 / 3],
              [-1 / 3, 1 / 3, -1 / 3],
              [-16 / 15, -8 / 15, -4 / 15],
              [1 / 15, -2 / 15, 4 / 15],
              [0, 0, 1]], np.float64)
# backward-filter, F(3, 4) on the same points: integer-scaled G' and A'^T with the inverse scales
G4 = np.array([[1, 0, 0, 0],
               [1, 1, 1, 1],
               [-1, 1, -1, 1],
               [-16, -8, -4, -2],
               [1, -2, 4, -8],
               [0, 0, 0, 1]], np.float64)
WGRAD_AT = np.array([[1, 1 / 3, 1 / 3, 1 / 15, 1 / 15, 0],
                     [0, 1 / 3, -1 / 3, 1 / 30, -2 / 15, 0],
                     [0, 1 / 3, 1 / 3, 1 / 60, 4 / 15, 1]], np.float64)


def _solve(rows, rhs):
    """Exact solution of a consistent (possibly over-determined) rational system."""
    n = len(rows[0])
    M = [list(r) + [b] for r, b in zip(rows, rhs)]
    row = 0
    for col in range(n):
        p = next(r for r in range(row, len(M)) if M[r][col] != 0)
        M[row], M[p] = M[p], M[row]
        pv = M[row][col]
        M[row] = [v / pv for v in M[row]]
        for r in range(len(M)):
            if r != row and M[r][col] != 0:
                f = M[r][col]
                M[r] = [a - f * b for a, b in zip(M[r], M[row])]
        row += 1
    assert all(M[r][n] == 0 for r in range(row, len(M))), 'inconsistent system'
    return [M[i][n] for i in range(n)]


def cook_toom(m, r, points=POINTS, bt=None):
    """(A^T [m x a], G [a x r], B^T [a x a]), a = m + r - 1, for the 1-D correlation
    y_i = sum_k g_k d_{i+k}:  y = A^T [(G g) * (B^T d)].  With ``bt`` given, G is solved for that
    B^T instead (F(3,4) shares B^T with F(4,3): the forward's transformed input is reused)."""
    a = m + r - 1
    p = list(points)
    assert len(p) == a - 1
    at = [[Fr(0)] * a for _ in range(m)]
    for j in range(a - 1):
        for i in range(m):
            at[i][j] = p[j] ** i
    at[m - 1][a - 1] = Fr(1)
    if bt is None:
        g = [[Fr(0)] * r for _ in range(a)]
        for j in range(a - 1):
            nj = Fr(1)
            for l in range(a - 1):
                if l != j:
                    nj *= p[j] - p[l]
            for k in range(r):
                g[j][k] = p[j] ** k / nj
        g[a - 1][r - 1] = Fr(1)
        bt = [[Fr(0)] * a for _ in range(a)]
        for q in range(a):
            rows, rhs = [], []
            for k in range(r):
                for i in range(m):
                    rows.append([at[i][x] * g[x][k] for x in range(a)])
                    rhs.append(Fr(1) if i + k == q else Fr(0))
            sol = _solve(rows, rhs)
            for x in range(a):
                bt[x][q] = sol[x]
    else:
        bt = [[Fr(v).limit_denominator(1 << 20) for v in row] for row in bt]
        g = [[Fr(0)] * r for _ in range(a)]
        for k in range(r):
            rows, rhs = [], []
            for q in range(a):
                for i in range(m):
                    rows.append([at[i][x] * bt[x][q] for x in range(a)])
                    rhs.append(Fr(1) if i + k == q else Fr(0))
            sol = _solve(rows, rhs)
            for x in range(a):
                g[x][k] = sol[x]
    f = lambda M: np.array([[float(v) for v in row] for row in M], np.float64)
    return f(at), f(g), f(bt)


def _tiles(x, size, step, off):
    """(N, C, H, W) -> (N, TH, TW, C, size, size) patches at (step*t + off), zero outside."""
    N, C, H, W = x.shape
    TH, TW = -(-H // 4), -(-W // 4)
    xp = np.zeros((N, C, 4 * TH + 2, 4 * TW + 2), x.dtype)
    xp[:, :, 1:1 + H, 1:1 + W] = x
    out = np.empty((N, TH, TW, C, size, size), x.dtype)
    for ty in range(TH):
        for tx in range(TW):
            y0, x0 = step * ty + off + 1, step * tx + off + 1
            out[:, ty, tx] = xp[:, :, y0:y0 + size, x0:x0 + size]
    return out


def conv3x3_fwd(x, w):
    """x (N,C,H,W), w (K,C,3,3), stride 1, pad 1 -> (N,K,H,W)."""
    N, C, H, W = x.shape
    K = w.shape[0]
    U = np.einsum('ai,kcij,bj->abkc', G, w, G)
    t = _tiles(x, 6, 4, -1)
    V = np.einsum('ai,nyxcij,bj->abnyxc', BT, t, BT)
    M = np.einsum('abnyxc,abkc->abnyxk', V, U)
    Y = np.einsum('ia,abnyxk,jb->nkyixj', AT, M, AT)
    TH, TW = t.shape[1], t.shape[2]
    return Y.reshape(N, K, 4 * TH, 4 * TW)[:, :, :H, :W]


def conv3x3_dgrad(g, w):
    """g (N,K,H,W) -> gx (N,C,H,W): the same pipeline with the flipped, transposed filter."""
    return conv3x3_fwd(g, np.ascontiguousarray(w[:, :, ::-1, ::-1].transpose(1, 0, 2, 3)))


def conv3x3_wgrad(x, g):
    """gw (K,C,3,3) from x (N,C,H,W) and g (N,K,H,W)."""
    V = np.einsum('ai,nyxcij,bj->abnyxc', BT, _tiles(x, 6, 4, -1), BT)
    Gy = np.einsum('ai,nyxkij,bj->abnyxk', G4, _tiles(g, 4, 4, 0), G4)
    dU = np.einsum('abnyxk,abnyxc->abkc', Gy, V)
    return np.einsum('ra,abkc,sb->kcrs', WGRAD_AT, dU, WGRAD_AT)
