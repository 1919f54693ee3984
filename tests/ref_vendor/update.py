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


# Round 4: the reference's OWN acceptance script of the extension, ops/test.py, and the autograd Function it imports
# (ops/functions/ms_deform_attn_func.py: forward AND backward through the compiled module), both whole and unmodified, so that
# tests/test_gpu_boundary.py can execute the script's three checks as written (fp64 forward equality, fp32 forward allclose,
# torch.autograd.gradcheck in double for D in {30, 32, 64, 71, 1025, 2048, 3096}) on top of the drop-in module.
OPS = "/root/reference/dinounet/dinov3/eval/segmentation/models/utils/ops"
WHOLE = [(os.path.join(OPS, "test.py"), os.path.join(HERE, "ops_test_ref.py")),
         (os.path.join(OPS, "functions", "ms_deform_attn_func.py"), os.path.join(HERE, "functions", "ms_deform_attn_func.py")),
         (os.path.join(OPS, "functions", "__init__.py"), os.path.join(HERE, "functions", "__init__.py"))]
WHOLE_HEADER = ("# TEST FIXTURE -- verbatim copy of the reference's %s (tests/ref_vendor/update.py); Copyright (c) Meta Platforms, Inc. and\n"
                "# affiliates / SenseTime (Deformable DETR, Apache-2.0), notices below.  Not product code: executed only by tests/test_gpu_boundary.py.\n")


def whole(src):
    rel = src[len("/root/reference/"):]
    return WHOLE_HEADER % rel + open(src).read()


if __name__ == "__main__":
    text = excerpt()
    if "--check" in sys.argv:
        ok = open(OUT).read() == text
        for src, dst in WHOLE:
            ok = ok and os.path.exists(dst) and open(dst).read() == whole(src)
        sys.exit(0 if ok else 1)
    open(OUT, "w").write(text)
    print(OUT)
    for src, dst in WHOLE:
        os.makedirs(os.path.dirname(dst), exist_ok=True)
        open(dst, "w").write(whole(src))
        print(dst)
