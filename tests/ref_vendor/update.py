"""TEST INFRASTRUCTURE ONLY.  Re-extracts tests/ref_vendor/ms_deform_attn_ref.py from the reference tree (build container only:
/root/reference does not exist on the GPU box).

What is vendored and why: lines 28-92 of dinounet/dinov3/eval/segmentation/models/utils/ms_deform_attn.py -- the reference's own
`MSDeformAttnFunction` (custom_fwd(cast_inputs=fp32) forward through `ms_deform_attn_core_pytorch`, once_differentiable backward through
`MSDA.ms_deform_attn_backward`) and `ms_deform_attn_core_pytorch` -- UNMODIFIED, so that tests/test_gpu_boundary.py can run the
reference's autograd Function on the GPU box on top of the drop-in `MultiScaleDeformableAttention` module (INTEGRATION.md section 2;
requested by VERDICT r2 "next round" item 1c).  It is a fixture of the boundary test, never imported by the product.

    python tests/ref_vendor/update.py          # rewrite the excerpt
    python tests/ref_vendor/update.py --check  # exit 1 if the committed excerpt differs from the reference's lines
"""
import os
import sys

SRC = "/root/reference/dinounet/dinov3/eval/segmentation/models/utils/ms_deform_attn.py"
FIRST, LAST = 28, 92
HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "ms_deform_attn_ref.py")
HEADER = '''# TEST FIXTURE -- verbatim excerpt (lines %d-%d) of the reference's
# dinounet/dinov3/eval/segmentation/models/utils/ms_deform_attn.py, extracted by tests/ref_vendor/update.py.
# Copyright (c) Meta Platforms, Inc. and affiliates; used and distributed under the terms of the DINOv3 License Agreement
# (the notice of the original file).  Not product code: imported only by tests/test_gpu_boundary.py to run the reference's own
# autograd Function on top of the drop-in MultiScaleDeformableAttention module.  The import block below restates what the
# excerpt needs from the original file's lines 6-18.
import torch
import torch.nn.functional as F
from torch.autograd import Function
from torch.amp import custom_fwd, custom_bwd
from torch.autograd.function import once_differentiable

import MultiScaleDeformableAttention as MSDA

# ---- verbatim from here ----
''' % (FIRST, LAST)


def excerpt():
    with open(SRC) as f:
        lines = f.readlines()
    return HEADER + "".join(lines[FIRST - 1:LAST])


if __name__ == "__main__":
    text = excerpt()
    if "--check" in sys.argv:
        sys.exit(0 if open(OUT).read() == text else 1)
    open(OUT, "w").write(text)
    print(OUT)
