"""TEST INFRASTRUCTURE ONLY -- deterministic synthetic weights for parity tests and the benchmark.

`make_state_dict(keys_shapes, seed)` fills every tensor of a DinoUNet state_dict from a per-key seeded
CPU generator (crc32 of the canonical key), so the build container (where the reference produced the
golden outputs) and the GPU box (where the HIP path is checked) see bit-identical weights without
shipping 100 MB+ checkpoints.  Unlike the reference's default init (sampling_offsets / attention_weights
zero, LayerScale 1e-5 -- MSA:138,151, HUB:225) every branch is numerically alive, so a wrong kernel shows.
Aliased keys (`decoder.encoder.*` DT:549, third-party `all_modules.N`) canonicalise to the same tensor.
"""
import math
import zlib

import torch


def canonical_key(k: str) -> str:
    if k.startswith("decoder.encoder."):
        k = k[len("decoder."):]
    return k.replace(".all_modules.0.", ".conv.").replace(".all_modules.1.", ".norm.")


def _gen(key: str, seed: int) -> torch.Generator:
    g = torch.Generator(device="cpu")
    g.manual_seed((zlib.crc32(key.encode()) ^ (seed * 0x9E3779B1)) & 0x7FFFFFFF)
    return g


def make_tensor(key: str, shape, dtype=torch.float32, seed: int = 0) -> torch.Tensor:
    key = canonical_key(key)
    shape = tuple(shape)
    g = _gen(key, seed)
    leaf = key.rsplit(".", 1)[-1]
    if leaf == "num_batches_tracked":
        return torch.zeros(shape, dtype=torch.int64)
    if leaf == "bias_mask":                       # LAY/attention.py:30-40: K third masked
        o = shape[0] // 3
        m = torch.ones(shape)
        m[o:2 * o] = 0
        return m
    if leaf == "periods":                         # LAY/rope_position_encoding.py:108-121, base 100
        n = shape[0]
        return 100.0 ** (2 * torch.arange(n, dtype=torch.float32) / (2 * n))
    if leaf == "running_var":
        return 0.5 + torch.rand(shape, generator=g)
    if leaf == "running_mean":
        return 0.1 * torch.randn(shape, generator=g)
    if leaf == "gamma":                           # LayerScale
        return 0.2 + 0.8 * torch.rand(shape, generator=g)
    if leaf in ("cls_token", "storage_tokens", "mask_token"):
        return 0.5 * torch.randn(shape, generator=g)
    if leaf == "level_embed":
        return torch.randn(shape, generator=g)
    if leaf == "bias":
        scale = 2.0 if key.endswith("sampling_offsets.bias") else 0.1
        return scale * torch.randn(shape, generator=g)
    if leaf == "weight":
        if len(shape) == 1:                       # norm scale
            return 1.0 + 0.1 * torch.randn(shape, generator=g)
        if ".up." in key or "up2." in key or "transpconvs." in key:   # ConvTranspose (Cin,Cout,kh,kw)
            fan_in = shape[0]
        else:
            fan_in = int(math.prod(shape[1:]))
        return torch.randn(shape, generator=g) / math.sqrt(fan_in)
    raise KeyError(f"no synthetic rule for state_dict key {key!r} shape {shape}")


def make_state_dict(keys_shapes, seed: int = 0):
    """keys_shapes: iterable of (key, shape).  Aliased keys share storage like the reference's do."""
    cache, out = {}, {}
    for k, shape in keys_shapes:
        ck = canonical_key(k)
        if ck not in cache:
            cache[ck] = make_tensor(ck, shape, seed=seed)
        assert tuple(cache[ck].shape) == tuple(shape), (k, shape, cache[ck].shape)
        out[k] = cache[ck]
    return out


def make_input(batch, channels, H, W, seed=0):
    g = torch.Generator(device="cpu"); g.manual_seed(1000 + seed)
    return torch.randn(batch, channels, H, W, generator=g)


def make_target(batch, H, W, num_classes, seed=0):
    g = torch.Generator(device="cpu"); g.manual_seed(2000 + seed)
    return torch.randint(0, num_classes, (batch, 1, H, W), generator=g)


def pinned_randomness(depth, batch, n_drop_paths=6, drop_prob=0.3, rescale=2.0, seed=0):
    """Fixed draws for the train-mode random ops, fed to BOTH sides of a parity test: one log-uniform RoPE rescale factor per ViT
    block (LAY/rope_position_encoding.py:93-97, redrawn in every block, VIT:271-272) and one per-sample DropPath mask per extractor
    (ADP:18-26, already divided by keep_prob).  Returns (log_scales (depth,) fp32, [masks (batch,) fp32] * n_drop_paths)."""
    g = torch.Generator(device="cpu"); g.manual_seed(3000 + seed)
    mx = math.log(rescale)
    log_scales = (torch.rand(depth, generator=g) * 2 - 1) * mx
    keep = 1.0 - drop_prob
    masks = [(torch.rand(batch, generator=g) < keep).float() / keep for _ in range(n_drop_paths)]
    masks[1][0] = 0.0            # make sure at least one sample is dropped and one kept somewhere
    masks[1][-1] = 1.0 / keep
    return log_scales, masks


def sample_indices(key: str, numel: int, n: int) -> torch.Tensor:
    """Deterministic element sample of a flattened tensor (gradient fixtures of tensors too large to commit in full)."""
    g = _gen("sample:" + canonical_key(key), 0)
    return torch.randperm(numel, generator=g)[:n].sort().values


def pinned_subsets(depth, batch, drop_rate, seed=0):
    """Fixed sample subsets for the ViT blocks' batch-subset stochastic depth (LAY/block.py:126-187): per block the first k entries of
    two random permutations of the batch (attention branch, FFN branch), k = max(int(B (1 - rate)), 1)."""
    g = torch.Generator(device="cpu"); g.manual_seed(4000 + seed)
    k = max(int(batch * (1 - drop_rate)), 1)
    return [(torch.randperm(batch, generator=g)[:k].clone(), torch.randperm(batch, generator=g)[:k].clone()) for _ in range(depth)]
