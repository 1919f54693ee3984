"""Generates tests/golden/sliding_window.npz from the REFERENCE's own helpers (run in the build container, where /root/reference
exists): compute_gaussian and compute_steps_for_sliding_window of dinounet/inference/sliding_window_prediction.py, imported as they
are (the module's only missing import, acvl_utils.pad_nd_image, is not used by these two functions and is stubbed).
usage: python oracle/make_golden_sw.py"""
import importlib.util
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference/dinounet/inference/sliding_window_prediction.py"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "sliding_window.npz")

STEP_CASES = [((512, 512), (512, 512), 0.5), ((600, 777), (512, 512), 0.5), ((110, 64), (64, 64), 0.5), ((1024, 1300), (512, 512), 0.5),
              ((160, 200), (128, 128), 0.5), ((300, 300), (128, 96), 0.25), ((129, 128), (128, 128), 1.0)]
GAUSS_CASES = [((64, 48), 1.0 / 8, 10.0), ((128, 128), 1.0 / 8, 10.0), ((33, 20), 1.0 / 8, 1.0)]


def main():
    for name in ("acvl_utils", "acvl_utils.cropping_and_padding", "acvl_utils.cropping_and_padding.padding"):
        sys.modules.setdefault(name, types.ModuleType(name))
    sys.modules["acvl_utils.cropping_and_padding.padding"].pad_nd_image = None
    spec = importlib.util.spec_from_file_location("ref_swp", REF)
    ref = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref)
    out = {}
    for i, (img, tile, step) in enumerate(STEP_CASES):
        st = ref.compute_steps_for_sliding_window(img, tile, step)
        out[f"steps{i}_args"] = np.array([*img, *tile, step], dtype=np.float64)
        for ax, v in enumerate(st):
            out[f"steps{i}_ax{ax}"] = np.array(v, dtype=np.int64)
    for i, (tile, sig, val) in enumerate(GAUSS_CASES):
        g = ref.compute_gaussian(tuple(tile), sigma_scale=sig, value_scaling_factor=val, dtype=torch.float32, device=torch.device("cpu"))
        out[f"gauss{i}_args"] = np.array([*tile, sig, val], dtype=np.float64)
        out[f"gauss{i}"] = g.numpy()
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, {k: v.shape for k, v in out.items() if not k.endswith("args")})


if __name__ == "__main__":
    main()
