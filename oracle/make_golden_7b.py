"""TEST INFRASTRUCTURE ONLY -- tests/golden/dinounet_7b_d4_64_eval.npz: the REFERENCE'S OWN `DinoUNet('dinounet_7b')` end to end at the 7B
WIDTH (embed dim 4096, 32 heads x 128, SwiGLU-64 FFN hidden 8192, no qkv bias, untied norms -- hub/backbones.py:452-496; adapter at D = 4096:
MSDeformAttn head width 128, ConvTranspose 4096 -> 4096, FAPM 4096 -> 256) with the backbone cut to depth 4 (interaction indexes
[0, 1, 2, 3]: get_intermediate_layers needs four distinct blocks) so that the weights (0.9 G parameters) and the CPU run stay small.  Eval mode, fp32, 64 x 64 input, batch 2.

Run in the build container only:   python -m oracle.make_golden_7b"""
import json
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
from oracle import refshim, weights  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
DEPTH, INDEXES = 4, [0, 1, 2, 3]


def main():
    torch.set_num_threads(8)
    refshim.install()
    import dinounet_training as DT
    DT.DINOv3_INTERACTION_INDEXES["dinounet_7b"] = list(INDEXES)
    import dinounet.dinov3.hub.backbones as HB
    make = HB._make_dinov3_vit

    def shallow(**kw):                    # the reference's dinov3_vit7b16 factory runs as is; only the depth it passes on is cut
        kw["depth"] = DEPTH
        return make(**kw)

    HB._make_dinov3_vit = shallow
    t0 = time.time()
    net = refshim.build_reference_dinounet("dinounet_7b", num_classes=2)
    ks = [(k, tuple(v.shape)) for k, v in net.state_dict().items()]
    nparam = sum(int(np.prod(s)) for k, s in ks if not k.startswith("decoder.encoder."))
    net.load_state_dict(weights.make_state_dict(ks, seed=0), strict=True)
    net.eval()
    B, H, W = 2, 64, 64
    x = weights.make_input(B, 3, H, W, seed=0)
    with torch.no_grad():
        y = net(x)
    print(f"[dinounet_7b depth {DEPTH}] {len(ks)} state-dict keys, {nparam / 1e6:.0f} M elements, forward done in {time.time() - t0:.0f} s; "
          f"logits {tuple(y.shape)} max |y| {float(y.abs().max()):.4f}")
    np.savez_compressed(os.path.join(GOLD, "dinounet_7b_d4_64_eval.npz"), logits=y.numpy().astype(np.float32),
                        meta=np.array(json.dumps(dict(model="dinounet_7b", depth=DEPTH, interaction_indexes=INDEXES, B=B, C=3, H=H, W=W,
                                                      num_classes=2, n_keys=len(ks)))))


if __name__ == "__main__":
    main()
