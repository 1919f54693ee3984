"""The nnU-Net 2D plans entry the reference hands to `DinoUNet.from_config` (dinounet_training.py:817): what
`dinounet/experiment_planning/experiment_planners/default_experiment_planner.py:377-396` emits for a 2D, 4-stage configuration
(key set consumed at dinounet/run/api.py:91-108).  Used by bench.py, the tools and the tests to build the benchmark network."""

PLANS_2D = {  # what default_experiment_planner.py:377-396 emits for 2d / 4 stages (API:91-108 key set)
    "architecture": {
        "network_class_name": "dynamic_network_architectures.architectures.unet.PlainConvUNet",
        "n_stages": 4, "features_per_stage": [32, 64, 128, 256],
        "kernel_sizes": [[3, 3]] * 4, "strides": [[1, 1], [2, 2], [2, 2], [2, 2]],
        "n_conv_per_stage": [2, 2, 2, 2], "n_conv_per_stage_decoder": [2, 2, 2],
        "conv_op": "torch.nn.modules.conv.Conv2d", "norm_op": "torch.nn.modules.instancenorm.InstanceNorm2d",
        "nonlin": "torch.nn.LeakyReLU", "conv_bias": True, "dropout_op": None,
        "norm_op_kwargs": {"eps": 1e-5, "affine": True}, "nonlin_kwargs": {"inplace": True},
        "dropout_op_kwargs": None,
    },
    "data_config": {"batch_size": 16, "patch_size": [512, 512]},
}
