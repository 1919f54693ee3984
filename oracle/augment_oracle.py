"""TEST INFRASTRUCTURE ONLY -- numpy / scipy restatement of the augmentation transforms behind dinounet_amd/augment.py.

The reference trainer (dinounet/training/nnUNetTrainer/nnUNetTrainer.py:684-776) builds its pipeline from the `batchgenerators` package
(requirements.txt), which is NOT vendored in /root/reference and not installed here: PARITY UNPINNED.  Each function restates the
algorithm batchgenerators 0.25 publishes for the transform named in its docstring, with the parameter draws passed in explicitly (the
product draws them the same way, dinounet_amd/augment.py: GPUAugment2D.draw).  `spatial` restates the product's Keys-bicubic image
resampling; `spatial_scipy_order3` is what batchgenerators itself calls (scipy map_coordinates, cubic B-spline) -- the tests report the
difference between the two on smooth data."""
import numpy as np
from scipy import ndimage


def _keys(t):
    t = np.abs(t)
    return np.where(t <= 1, (1.5 * t - 2.5) * t * t + 1, np.where(t < 2, ((-0.5 * t + 2.5) * t - 4) * t + 2, 0.0))


def _coords(prm, Hi, Wi, Ho, Wo):
    """batchgenerators.augmentations.utils: create_zero_centered_coordinate_mesh, rotate_coords_2d, scale_coords, + image centre;
    MirrorTransform (flip of the output arrays) folded in"""
    ys, xs = np.meshgrid(np.arange(Ho, dtype=np.float64), np.arange(Wo, dtype=np.float64), indexing="ij")
    if prm[4]:
        ys = Ho - 1 - ys
    if prm[5]:
        xs = Wo - 1 - xs
    cy, cx = ys - 0.5 * (Ho - 1), xs - 0.5 * (Wo - 1)
    y = prm[0] * cy + prm[1] * cx + (Hi / 2.0 - 0.5)
    x = prm[2] * cy + prm[3] * cx + (Wi / 2.0 - 0.5)
    return y, x


def _bicubic(img, y, x):
    H, W = img.shape
    y0, x0 = np.floor(y).astype(int), np.floor(x).astype(int)
    out = np.zeros_like(y)
    for j in range(-1, 3):
        for i in range(-1, 3):
            yy, xx = y0 + j, x0 + i
            ok = (yy >= 0) & (yy < H) & (xx >= 0) & (xx < W)
            v = np.where(ok, img[np.clip(yy, 0, H - 1), np.clip(xx, 0, W - 1)], 0.0)
            out += _keys(y - yy) * _keys(x - xx) * v
    return out


def spatial(data, seg, prm, patch):
    """SpatialTransform + MirrorTransform: data Keys-bicubic (zero outside), seg by interpolate_img's order-1 rule
    (`for c in unique(seg): res = map_coordinates(seg == c, order=1, cval); result[res >= 0.5] = c`, out-of-image -> 0)"""
    B, C, Hi, Wi = data.shape
    Ho, Wo = patch
    out = np.zeros((B, C, Ho, Wo), np.float64)
    sout = None if seg is None else np.zeros((B, 1, Ho, Wo), np.float64)
    for b in range(B):
        y, x = _coords(prm[b], Hi, Wi, Ho, Wo)
        for c in range(C):
            out[b, c] = _bicubic(data[b, c].astype(np.float64), y, x)
        if seg is not None:
            res = np.zeros((Ho, Wo))
            for lab in np.sort(np.unique(seg[b, 0])):
                m = ndimage.map_coordinates((seg[b, 0] == lab).astype(np.float64), [y, x], order=1, mode="constant", cval=-1.0)
                res[m >= 0.5] = lab
            sout[b, 0] = res
    return out, sout


def spatial_scipy_order3(data, prm, patch):
    B, C, Hi, Wi = data.shape
    out = np.zeros((B, C) + tuple(patch))
    for b in range(B):
        y, x = _coords(prm[b], Hi, Wi, *patch)
        for c in range(C):
            out[b, c] = ndimage.map_coordinates(data[b, c].astype(np.float64), [y, x], order=3, mode="constant", cval=0.0)
    return out


def contrast(x, factor):
    """augment_contrast(preserve_range=True, per_channel=True): (x - mean) * f + mean, clipped to the plane's old [min, max]"""
    out = x.copy()
    for p in np.ndindex(x.shape[:2]):
        if factor[p] != 1:
            mn, lo, hi = x[p].mean(), x[p].min(), x[p].max()
            out[p] = np.clip((x[p] - mn) * factor[p] + mn, lo, hi)
    return out


def gamma(x, g, invert, retain_stats=True):
    """augment_gamma(per_channel=True, retain_stats=True, epsilon=1e-7)"""
    out = x.copy()
    for p in np.ndindex(x.shape[:2]):
        if g[p] <= 0:
            continue
        v = -x[p] if invert else x[p].copy()
        mn, sd = v.mean(), v.std()
        lo = v.min()
        rng = v.max() - lo
        v = np.power((v - lo) / (rng + 1e-7), g[p]) * rng + lo
        if retain_stats:
            v = (v - v.mean()) / (v.std() + 1e-8) * sd + mn
        out[p] = -v if invert else v
    return out


def blur(x, sigma):
    """augment_gaussian_blur: scipy.ndimage.gaussian_filter(plane, sigma, order=0) (mode 'reflect', truncate 4)"""
    out = x.copy()
    for p in np.ndindex(x.shape[:2]):
        if sigma[p] > 0:
            out[p] = ndimage.gaussian_filter(x[p].astype(np.float64), float(sigma[p]), order=0)
    return out


def lowres(x, zoom):
    """augment_linear_downsampling_scipy(order_downsample=0, order_upsample=3) with skimage.transform.resize semantics restated:
    nearest source pixel floor((i + 0.5) H / Hl) going down, Keys cubic at (y + 0.5) Hl / H - 0.5 with replicated edges going up"""
    out = x.copy()
    H, W = x.shape[2:]
    for p in np.ndindex(x.shape[:2]):
        z = zoom[p]
        if not (0 < z < 1):
            continue
        Hl, Wl = max(int(np.rint(H * z)), 1), max(int(np.rint(W * z)), 1)
        ys = np.minimum(((np.arange(Hl) + 0.5) * H / Hl).astype(int), H - 1)
        xs = np.minimum(((np.arange(Wl) + 0.5) * W / Wl).astype(int), W - 1)
        low = x[p][np.ix_(ys, xs)].astype(np.float64)
        fy = (np.arange(H) + 0.5) * Hl / H - 0.5
        fx = (np.arange(W) + 0.5) * Wl / W - 0.5
        y0, x0 = np.floor(fy).astype(int), np.floor(fx).astype(int)
        acc = np.zeros((H, W))
        for j in range(-1, 3):
            wy = _keys(fy - (y0 + j))[:, None]
            yl = np.clip(y0 + j, 0, Hl - 1)
            for k in range(-1, 3):
                wx = _keys(fx - (x0 + k))[None, :]
                xl = np.clip(x0 + k, 0, Wl - 1)
                acc += wy * wx * low[np.ix_(yl, xl)]
        out[p] = acc
    return out
