"""GPU-side training augmentation for 2D slices.

`GPUAugment2D` applies the transform list of the reference trainer -- `nnUNetTrainer.get_training_transforms`
(dinounet/training/nnUNetTrainer/nnUNetTrainer.py:684-776) with the 2D configuration of
`configure_rotation_dummyDA_mirroring_and_inital_patch_size` (:391-441) -- to a batch that is already on the MI355X, through the HIP
kernels of csrc/augment.hip (C ABI `du_aug_*`).  The random decisions (which sample gets which transform, with which parameter) are
drawn on the host from a numpy RandomState exactly as batchgenerators does it (`np.random.uniform() < p_per_sample`, uniform parameter
ranges); only the per-pixel work runs on the GPU.  The reference hands the transforms to the `batchgenerators` package, which is not
vendored in the reference tree: parity is unpinned against it (tests compare the kernels with a numpy / scipy restatement,
oracle/augment_oracle.py).  Order and probabilities:

    SpatialTransform   rotation U(-180deg, 180deg) (|angle| <= 15deg for patches with aspect > 1.5), p 0.2; scale p 0.2, log-symmetric
                       draw from (0.7, 1.4) as batchgenerators does (U(0.7,1) or U(1,1.4) with probability 1/2 each); centre crop
    GaussianNoise      p 0.1, variance U(0, 0.1)
    GaussianBlur       p 0.2 per sample, p 0.5 per channel, sigma U(0.5, 1) per channel
    BrightnessMult     p 0.15, multiplier U(0.75, 1.25) per channel (per_channel=True is batchgenerators' default)
    Contrast           p 0.15, factor from (0.75, 1.25) (U(0.75,1) or U(1,1.25)), preserve_range, per channel
    SimulateLowRes     p 0.25 per sample, p 0.5 per channel, zoom U(0.5, 1)
    Gamma (inverted)   p 0.1, gamma from (0.7, 1.5), retain_stats
    Gamma              p 0.3, same
    Mirror             each axis with p 0.5
    RemoveLabel(-1,0)  (out-of-image labels are already 0)
"""
import ctypes as C

import numpy as np
import torch

from . import _lib


def _p(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def _st():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


class GPUAugment2D:
    def __init__(self, patch_size, seed=None):
        self.patch = tuple(int(v) for v in patch_size)
        assert len(self.patch) == 2
        lim = 15.0 if max(self.patch) / min(self.patch) > 1.5 else 180.0            # nnUNetTrainer.py:401-412
        self.angle = (-lim / 360 * 2 * np.pi, lim / 360 * 2 * np.pi)
        self.rs = np.random.RandomState(seed)

    # ---- parameter draws (host) ---------------------------------------------------------------------------------------------
    def _two_sided(self, lo, hi):
        """batchgenerators' draw for ranges straddling 1 (scale, contrast, gamma): below / above 1 with probability 1/2"""
        if self.rs.random_sample() < 0.5 and lo < 1:
            return self.rs.uniform(lo, 1)
        return self.rs.uniform(max(lo, 1), hi)

    def draw(self, B, Cc):
        """All random decisions of one batch -> dict of numpy arrays (also the input of the oracle restatement)."""
        rs = self.rs
        prm = np.zeros((B, 6), np.float32)
        for b in range(B):
            ang = rs.uniform(*self.angle) if rs.uniform() < 0.2 else 0.0
            sc = self._two_sided(0.7, 1.4) if rs.uniform() < 0.2 else 1.0
            c, s = np.cos(ang), np.sin(ang)
            prm[b, :4] = np.array([c, -s, s, c]) * sc          # coords = R(angle) @ coords, then * scale (rotate_coords_2d, scale_coords)
        noise = np.zeros((B, Cc), np.float32)
        for b in range(B):
            if rs.uniform() < 0.1:
                # batchgenerators' augment_gaussian_noise draws `variance = uniform(0, 0.1)` and hands it to np.random.normal as the
                # STANDARD DEVIATION (scale): the noise std is U(0, 0.1), not its square root (per_channel=False: one draw per sample)
                noise[b] = rs.uniform(0, 0.1)
        blur = np.zeros((B, Cc), np.float32)
        for b in range(B):
            if rs.uniform() < 0.2:
                for c_ in range(Cc):
                    if rs.uniform() <= 0.5:
                        blur[b, c_] = rs.uniform(0.5, 1.0)
        mult = np.ones((B, Cc), np.float32)
        for b in range(B):
            if rs.uniform() < 0.15:
                mult[b] = rs.uniform(0.75, 1.25, Cc)
        contrast = np.ones((B, Cc), np.float32)
        for b in range(B):
            if rs.uniform() < 0.15:
                contrast[b] = [self._two_sided(0.75, 1.25) for _ in range(Cc)]
        zoom = np.zeros((B, Cc), np.float32)
        for b in range(B):
            if rs.uniform() < 0.25:
                for c_ in range(Cc):
                    if rs.uniform() < 0.5:
                        zoom[b, c_] = rs.uniform(0.5, 1.0)
        gam = []
        for p_ in (0.1, 0.3):
            g = np.zeros((B, Cc), np.float32)
            for b in range(B):
                if rs.uniform() < p_:
                    g[b] = [self._two_sided(0.7, 1.5) for _ in range(Cc)]
            gam.append(g)
        for b in range(B):
            prm[b, 4] = float(rs.uniform() < 0.5)
            prm[b, 5] = float(rs.uniform() < 0.5)
        return dict(spatial=prm, noise_sigma=noise, blur_sigma=blur, mult=mult, contrast=contrast, zoom=zoom, gamma_inv=gam[0], gamma=gam[1],
                    seed=int(rs.randint(0, 2 ** 31 - 1)))

    # ---- device work --------------------------------------------------------------------------------------------------------
    def apply(self, data, seg, d):
        """data (B, C, Hi, Wi) fp32 cuda, seg (B, 1, Hi, Wi) fp32 labels (or None), d = draw(...) -> (data_aug (B,C,H,W), target (B,1,H,W))"""
        if not data.is_cuda:
            raise RuntimeError("GPUAugment2D runs on the MI355X through libdinounet_hip.so (no CPU fallback)")
        L = _lib.lib()
        B, Cc, Hi, Wi = data.shape
        H, W = self.patch
        dev = data.device
        dv = lambda a: torch.from_numpy(np.ascontiguousarray(a, np.float32)).to(dev)
        data = data.contiguous().float()
        seg = None if seg is None else seg.contiguous().float()
        out = torch.empty((B, Cc, H, W), dtype=torch.float32, device=dev)
        sout = None if seg is None else torch.empty((B, 1, H, W), dtype=torch.float32, device=dev)
        # every parameter array is a named device tensor that outlives its launch (a temporary handed to ctypes as a raw pointer would be
        # freed -- and its block re-used by the next upload -- before the kernel is even enqueued)
        ones = np.ones((B, Cc), np.float32)
        t_sp, t_noise, t_one, t_zero = dv(d["spatial"]), dv(d["noise_sigma"]), dv(ones), dv(0 * ones)
        t_blur, t_mult, t_con, t_zoom = dv(d["blur_sigma"]), dv(d["mult"]), dv(d["contrast"]), dv(d["zoom"])
        _lib.check(L.du_aug_spatial(_p(data), _p(seg), _p(t_sp), _p(out), _p(sout), B, Cc, Hi, Wi, H, W, _st()), "du_aug_spatial")
        P, n = B * Cc, H * W
        tmp = torch.empty_like(out)
        stats = torch.empty((P, 4), dtype=torch.float32, device=dev)
        # GaussianNoise (before blur and brightness in the reference order; the multiplier pass comes after the blur)
        _lib.check(L.du_aug_noise_mult(_p(out), _p(t_noise), _p(t_one), P, n, d["seed"], _st()), "du_aug_noise_mult")
        if (d["blur_sigma"] > 0).any():
            _lib.check(L.du_aug_blur(_p(out), _p(tmp), _p(t_blur), P, H, W, 0, _st()), "du_aug_blur")
            _lib.check(L.du_aug_blur(_p(tmp), _p(out), _p(t_blur), P, H, W, 1, _st()), "du_aug_blur")
        _lib.check(L.du_aug_noise_mult(_p(out), _p(t_zero), _p(t_mult), P, n, 0, _st()), "du_aug_noise_mult")
        if (d["contrast"] != 1).any():
            _lib.check(L.du_aug_plane_stats(_p(out), _p(stats), P, n, _st()), "du_aug_plane_stats")
            _lib.check(L.du_aug_contrast(_p(out), _p(t_con), _p(stats), P, n, _st()), "du_aug_contrast")
        if ((d["zoom"] > 0) & (d["zoom"] < 1)).any():
            _lib.check(L.du_aug_lowres(_p(out), _p(tmp), _p(t_zoom), P, H, W, _st()), "du_aug_lowres")
            out, tmp = tmp, out
        keep = []                      # operands of the asynchronous launches below stay referenced until apply() returns
        for key, inv in (("gamma_inv", 1.0), ("gamma", 0.0)):
            g = d[key]
            if not (g > 0).any():
                continue
            on = (g > 0).astype(np.float32)
            t_g, t_inv, onv = dv(g), dv(on * inv), dv(on).view(-1)
            _lib.check(L.du_aug_plane_stats(_p(out), _p(stats), P, n, _st()), "du_aug_plane_stats")       # mean / std to retain, range
            before = stats.clone()
            _lib.check(L.du_aug_gamma(_p(out), _p(t_g), _p(t_inv), _p(stats), P, n, _st()), "du_aug_gamma")
            _lib.check(L.du_aug_plane_stats(_p(out), _p(stats), P, n, _st()), "du_aug_plane_stats")
            # retain_stats on the (possibly negated) image: x = (x - mean') / (std' + 1e-8) * std + mean, then negate back
            sgn = 1.0 - 2.0 * t_inv.view(-1)
            mean_b, std_b = before[:, 0] * sgn, before[:, 1]
            a = std_b / (stats[:, 1] + 1e-8)
            bb = mean_b - stats[:, 0] * a
            a = ((a * sgn) * onv + (1 - onv)).contiguous()
            bb = ((bb * sgn) * onv).contiguous()
            _lib.check(L.du_aug_affine(_p(out), _p(a), _p(bb), P, n, _st()), "du_aug_affine")
            keep += [t_g, t_inv, onv, a, bb, before]
        return out, sout

    def __call__(self, data, seg=None):
        d = self.draw(data.shape[0], data.shape[1])
        return self.apply(data, seg, d)
