"""TEST INFRASTRUCTURE (oracle): CPU restatement of the reference's sliding-window inference with Gaussian blending.
Follows dinounet/inference/sliding_window_prediction.py:10-60 (compute_gaussian, compute_steps_for_sliding_window) and
dinounet/inference/predict_from_raw_data.py:503-535 (slicers), :571-621 (accumulate / normalise), :680-727 (pad, predict, un-pad).
Pinned: oracle/make_golden_sw.py imports the reference's own two helper functions in the build container and stores their outputs in
tests/golden/sliding_window.npz; tests/test_cpu_oracle_and_boundary.py checks this restatement (and the product's host logic)
against them.  `acvl_utils.pad_nd_image` is a third-party dependency that is not vendored (requirements: acvl-utils; no lock file):
restated from its published behaviour -- parity unpinned for the padding of images smaller than the patch.
Only tests/ may import this module."""
import numpy as np
import torch
from scipy.ndimage import gaussian_filter


def compute_gaussian(tile_size, sigma_scale=1.0 / 8, value_scaling_factor=1.0, dtype=torch.float32):
    tmp = np.zeros(tile_size)                                           # sliding_window_prediction.py:14-18
    tmp[tuple(i // 2 for i in tile_size)] = 1
    g = gaussian_filter(tmp, [i * sigma_scale for i in tile_size], 0, mode="constant", cval=0)
    g = torch.from_numpy(g)
    g = (g / torch.max(g) * value_scaling_factor).type(dtype)           # :22-23
    g[g == 0] = torch.min(g[g != 0])                                    # :26-27
    return g


def compute_steps(image_size, tile_size, tile_step_size):
    target = [i * tile_step_size for i in tile_size]                    # :41
    num_steps = [int(np.ceil((i - k) / j)) + 1 for i, j, k in zip(image_size, target, tile_size)]    # :43
    steps = []
    for dim in range(len(tile_size)):                                   # :46-58
        max_step_value = image_size[dim] - tile_size[dim]
        actual = max_step_value / (num_steps[dim] - 1) if num_steps[dim] > 1 else 99999999999
        steps.append([int(np.round(actual * i)) for i in range(num_steps[dim])])
    return steps


def pad_nd_image_2d(data, patch_size):
    """acvl_utils pad_nd_image(image, new_shape, 'constant', {'value': 0}, return_slicer=True) on the last two axes."""
    old = np.array(data.shape[-2:])
    new = np.maximum(old, np.array(patch_size))
    diff = new - old
    below, above = diff // 2, diff // 2 + diff % 2
    out = torch.zeros((*data.shape[:-2], int(new[0]), int(new[1])), dtype=data.dtype)
    out[..., below[0]:below[0] + old[0], below[1]:below[1] + old[1]] = data
    return out, (slice(int(below[0]), int(new[0] - above[0])), slice(int(below[1]), int(new[1] - above[1])))


def mirror_and_predict(predict, x, mirror_axes):
    """predict_from_raw_data.py:537-552 (_internal_maybe_mirror_and_predict): the prediction plus the flipped-back predictions of every
    non-empty combination of the allowed mirror axes (spatial axis m = tensor dim m + 2), divided by their number + 1."""
    import itertools
    prediction = predict(x)
    if mirror_axes is not None:
        assert max(mirror_axes) <= x.ndim - 3
        combos = [c for i in range(len(mirror_axes)) for c in itertools.combinations([m + 2 for m in mirror_axes], i + 1)]
        for axes in combos:
            prediction = prediction + torch.flip(predict(torch.flip(x, (*axes,))), (*axes,))
        prediction = prediction / (len(combos) + 1)
    return prediction


def predict_sliding_window_logits(predict, data, patch_size, tile_step_size=0.5, use_gaussian=True, accum_dtype=torch.float32,
                                  mirror_axes=None):
    """`predict(window (1, C, ph, pw)) -> logits (1, K, ph, pw)`; data (C, D, H, W).  One window per call, like the reference."""
    if mirror_axes is not None:
        inner = predict
        predict = lambda w: mirror_and_predict(inner, w, mirror_axes)
    data, (ys, xs) = pad_nd_image_2d(data, patch_size)                   # predict_from_raw_data.py:703-705
    D, H, W = data.shape[1:]
    steps = compute_steps((H, W), patch_size, tile_step_size)           # :512
    gauss = compute_gaussian(tuple(patch_size), 1.0 / 8, 10, accum_dtype) if use_gaussian else None     # :595-597
    pred = npred = None
    for d in range(D):                                                  # :517-522
        for sy in steps[0]:
            for sx in steps[1]:
                sl = (slice(None), d, slice(sy, sy + patch_size[0]), slice(sx, sx + patch_size[1]))
                p = predict(data[sl][None])[0].to(accum_dtype)          # :602-605
                if pred is None:
                    pred = torch.zeros((p.shape[0], D, H, W), dtype=accum_dtype)        # :590-593
                    npred = torch.zeros((D, H, W), dtype=accum_dtype)
                pred[sl] += p * gauss if use_gaussian else p            # :607
                npred[sl[1:]] += gauss if use_gaussian else 1           # :608
    pred /= npred                                                       # :610
    return pred[:, :, ys, xs]                                           # :726
