"""Generates tests/golden/sliding_window_mirror.npz from the REFERENCE's own test-time-mirroring routine (run in the build container, where
/root/reference exists): nnUNetPredictor._internal_maybe_mirror_and_predict of dinounet/inference/predict_from_raw_data.py:537-552, imported
as it is (its nnU-Net / batchgenerators / acvl_utils imports are not touched by this method and are stubbed) and called unbound on a
stand-in `self` carrying a small deterministic, non-symmetric "network" -- so the golden pins the flip combinations, their order of
accumulation and the normalisation, not a model.
usage: python oracle/make_golden_mirror.py"""
import importlib.abc
import importlib.machinery
import importlib.util
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference/dinounet/inference/predict_from_raw_data.py"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "sliding_window_mirror.npz")
STUB_ROOTS = ("dinounet", "batchgenerators", "acvl_utils", "nnunetv2", "dynamic_network_architectures", "tqdm", "SimpleITK", "nibabel", "skimage")
CASES = [(None,), ((0,),), ((1,),), ((0, 1),)]


def toy_network(x):
    """(b, 3, h, w) -> (b, 2, h, w): not flip-equivariant in either axis (shifts), not linear (square)"""
    return (x * x + 0.5 * x.roll(1, -1) + 0.25 * x.roll(1, -2))[:, :2] + 0.1 * x[:, 2:3]


class _Stub(types.ModuleType):
    def __getattr__(self, k):
        if k.startswith("__"):
            raise AttributeError(k)
        return type(k, (object,), {})


class _Finder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    def find_spec(self, name, path, target=None):
        return importlib.machinery.ModuleSpec(name, self, is_package=True) if name.split(".")[0] in STUB_ROOTS else None

    def create_module(self, spec):
        m = _Stub(spec.name)
        m.__path__ = []
        return m

    def exec_module(self, m):
        pass


def main():
    sys.meta_path.insert(0, _Finder())
    spec = importlib.util.spec_from_file_location("ref_pred", REF)
    ref = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref)
    g = torch.Generator().manual_seed(11)
    x = torch.randn(3, 3, 6, 10, generator=g)
    out = {"x": x.numpy()}
    for i, (axes,) in enumerate(CASES):
        me = types.SimpleNamespace(network=toy_network, allowed_mirroring_axes=axes, use_mirroring=axes is not None)
        y = ref.nnUNetPredictor._internal_maybe_mirror_and_predict(me, x.clone())
        out[f"axes{i}"] = np.array([-1] if axes is None else list(axes), dtype=np.int64)
        out[f"y{i}"] = y.numpy()
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, {k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
