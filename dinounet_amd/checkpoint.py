"""Checkpoint I/O that skips what never changes -- SURVEY.md 8(f) rank 3.

The reference writes `mod.state_dict()` into every checkpoint (dinounet/training/nnUNetTrainer/nnUNetTrainer.py:1083-1106: every 50
epochs and at every best-EMA epoch, :1067-1076).  For Dino U-Net that dict holds the frozen DINOv3 backbone (86 M / 304 M / 6.7 B
parameters that training never touches) and, because `UNetDecoder` registers the encoder as a sub-module (dinounet_training.py:549),
the WHOLE encoder a second time under `decoder.encoder.*` (471 of the 1 002 keys of dinounet_s).  The compact form stores
  * every entry that training can change: trainable parameters and buffers (BatchNorm running statistics, num_batches_tracked),
  * a fingerprint of the frozen backbone instead of its tensors,
and `expand_state_dict` rebuilds the exact dict the reference's `load_checkpoint` (:1108-1144, `load_state_dict(strict)`) expects:
frozen entries come from the live module (which loaded them from the DINOv3 .pth at construction, dinounet_training.py:51-70) or
from a backbone state dict passed in; `decoder.encoder.X` entries are the same tensors as `encoder.X`.

    save_checkpoint(path, net, optimizer, current_epoch=..., ...)        # compact file
    load_checkpoint(path, net, optimizer)                                 # strict load into a dinounet_amd or reference module
    to_reference_checkpoint(ckpt, net)                                    # dict accepted by nnUNetTrainer.load_checkpoint(dict)
Pure host code (no GPU arithmetic); tensors are written as they are (any device)."""
import hashlib

import torch

FORMAT = "dinounet_amd.compact/1"
ALIAS_PREFIX = "decoder.encoder."
ALIAS_OF = "encoder."
FROZEN_PREFIX = "encoder.dinov3_adapter.backbone."


def _fingerprint(items):
    """sha256 over (key, shape, dtype, raw bytes) of the frozen tensors, in key order."""
    h = hashlib.sha256()
    for k, v in sorted(items, key=lambda kv: kv[0]):
        t = v.detach().contiguous().cpu()
        h.update(k.encode())
        h.update(str((tuple(t.shape), str(t.dtype))).encode())
        h.update(t.view(torch.uint8).numpy().tobytes() if t.numel() else b"")
    return h.hexdigest()


def _frozen_keys(net):
    """state_dict keys under the backbone whose parameters do not require grad, plus the backbone's buffers (bias_mask, RoPE periods):
    everything `DINOv3_Adapter` freezes (dinov3_adapter.py:338-341)."""
    trainable = {FROZEN_PREFIX + n for n, p in net.encoder.dinov3_adapter.backbone.named_parameters() if p.requires_grad}
    return [k for k in net.state_dict().keys() if k.startswith(FROZEN_PREFIX) and k not in trainable]


def compact_state_dict(net, fingerprint=True):
    """-> (compact dict, meta).  Keys keep the reference's names, so the compact dict is a strict subset of `net.state_dict()`."""
    full = net.state_dict()
    frozen = set(_frozen_keys(net))
    compact = {k: v for k, v in full.items() if not k.startswith(ALIAS_PREFIX) and k not in frozen}
    fp = None
    if fingerprint:
        # hashing 1.2 GB (dinounet_l) costs ~1.3 s; the frozen tensors never change, so the digest is cached on the module and keyed by
        # their storage addresses and autograd version counters
        sig = tuple((k, full[k].data_ptr(), full[k]._version) for k in sorted(frozen))
        cached = net.__dict__.get("_frozen_fingerprint_cache")
        if cached is None or cached[0] != sig:
            cached = (sig, _fingerprint([(k, full[k]) for k in frozen]))
            net.__dict__["_frozen_fingerprint_cache"] = cached
        fp = cached[1]
    meta = {"format": FORMAT, "n_full_keys": len(full), "n_frozen_keys": len(frozen), "frozen_prefix": FROZEN_PREFIX,
            "alias_prefix": ALIAS_PREFIX, "alias_of": ALIAS_OF, "frozen_fingerprint": fp, "key_order": list(full.keys())}
    return compact, meta


def expand_state_dict(compact, meta, net=None, backbone_state=None, check=True):
    """Compact dict -> the full `network_weights` dict of the reference checkpoint (same keys, same order).  Frozen backbone entries come
    from `backbone_state` (a DINOv3 state dict keyed like `DinoVisionTransformer.state_dict()`, or already prefixed) or from `net`."""
    if meta.get("format") != FORMAT:
        raise ValueError(f"not a {FORMAT} checkpoint: {meta.get('format')!r}")
    fp, ap, ao = meta["frozen_prefix"], meta["alias_prefix"], meta["alias_of"]
    src = {}
    if backbone_state is not None:
        src = {(k if k.startswith(fp) else fp + k): v for k, v in backbone_state.items()}
    elif net is not None:
        src = {k: v for k, v in net.state_dict().items() if k.startswith(fp)}
    full = {}
    frozen_items = []
    for k in meta["key_order"]:
        if k.startswith(ap):
            continue
        if k in compact:
            full[k] = compact[k]
        elif k in src:
            full[k] = src[k]
            frozen_items.append((k, src[k]))
        else:
            raise KeyError(f"{k}: neither in the compact checkpoint nor in the supplied backbone weights")
    if check and meta.get("frozen_fingerprint") is not None:
        got = _fingerprint(frozen_items)
        if got != meta["frozen_fingerprint"]:
            raise ValueError("the frozen backbone supplied for expansion is not the one this checkpoint was trained on (fingerprint mismatch)")
    out = {}
    for k in meta["key_order"]:                      # the reference's key order; aliases share storage with their originals
        out[k] = full[ao + k[len(ap):]] if k.startswith(ap) else full[k]
    return out


def save_checkpoint(path, net, optimizer=None, grad_scaler=None, **trainer_fields):
    """nnUNetTrainer.save_checkpoint (:1083-1106) with compact `network_weights`.  trainer_fields: logging, _best_ema, current_epoch,
    init_args, trainer_name, inference_allowed_mirroring_axes -- stored verbatim."""
    compact, meta = compact_state_dict(net)
    ckpt = {"network_weights": compact, "network_weights_meta": meta,
            "optimizer_state": optimizer.state_dict() if optimizer is not None else None,
            "grad_scaler_state": grad_scaler.state_dict() if grad_scaler is not None else None}
    ckpt.update(trainer_fields)
    torch.save(ckpt, path)
    return ckpt


def to_reference_checkpoint(ckpt, net=None, backbone_state=None, check=True):
    """Compact checkpoint dict -> a dict `nnUNetTrainer.load_checkpoint` accepts as is (:1108: it takes a dict or a file name)."""
    if "network_weights_meta" not in ckpt:
        return ckpt
    out = {k: v for k, v in ckpt.items() if k != "network_weights_meta"}
    out["network_weights"] = expand_state_dict(ckpt["network_weights"], ckpt["network_weights_meta"], net, backbone_state, check)
    return out


def load_checkpoint(path_or_ckpt, net, optimizer=None, grad_scaler=None, map_location=None, check=True):
    """Strict load of a compact (or plain reference) checkpoint into `net`; returns the checkpoint dict (trainer fields included)."""
    ckpt = torch.load(path_or_ckpt, map_location=map_location, weights_only=False) if isinstance(path_or_ckpt, str) else path_or_ckpt
    ref = to_reference_checkpoint(ckpt, net, None, check)
    net.load_state_dict(ref["network_weights"], strict=True)
    if optimizer is not None and ref.get("optimizer_state") is not None:
        optimizer.load_state_dict(ref["optimizer_state"])
    if grad_scaler is not None and ref.get("grad_scaler_state") is not None:
        grad_scaler.load_state_dict(ref["grad_scaler_state"])
    return ref
