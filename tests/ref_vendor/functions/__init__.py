# TEST FIXTURE -- verbatim copy of the reference's dinounet/dinov3/eval/segmentation/models/utils/ops/functions/__init__.py (tests/ref_vendor/update.py); Copyright (c) Meta Platforms, Inc. and
# affiliates / SenseTime (Deformable DETR, Apache-2.0), notices below.  Not product code: executed only by tests/test_gpu_boundary.py.
# Copyright (c) Meta Platforms, Inc. and affiliates.
#
# This software may be used and distributed in accordance with
# the terms of the DINOv3 License Agreement.

# ------------------------------------------------------------------------------------------------
# Deformable DETR
# Copyright (c) 2020 SenseTime. All Rights Reserved.
# Licensed under the Apache License, Version 2.0 [see LICENSE for details]
# ------------------------------------------------------------------------------------------------
# Modified from https://github.com/chengdazhi/Deformable-Convolution-V2-PyTorch/tree/pytorch_1.0.0
# ------------------------------------------------------------------------------------------------

from .ms_deform_attn_func import MSDeformAttnFunction
