"""Module-level API of the reference's top-level script `dinounet_training.py` (DT), re-exported over the HIP
implementation: `DinoUNet` & friends, the model registries, and the nnU-Net trainer plug-ins whose static
`build_network_architecture` hook (nnUNetTrainer.py:266-271, override DT:857-881) is the drop-in boundary.

If the reference's nnU-Net fork (`dinounet.training...nnUNetTrainerNoDeepSupervision`) is importable, the trainer
classes derive from it exactly like DT:833 and `main_dinov3` drives its plan/preprocess/train/evaluate façade
(DT:958-1050, out of this repo's scope).  Otherwise they derive from a minimal stand-in so the plug-in surface stays
importable (this is the case on the GPU box and in CI, where the reference tree and batchgenerators are absent).
"""
from typing import List, Tuple, Union

from torch import nn

from .network_architecture.dinounet import (DINOv3_INTERACTION_INDEXES, DINOv3_MODEL_FACTORIES, DINOv3_MODEL_INFO,  # noqa: F401
                                            DepthwiseSeparableConv, DINOv3EncoderAdapter, DinoUNet, FAPM, LearnableUpsampleBlock,
                                            SqueezeExcitation, UNetDecoder, load_dinov3_model)

try:  # the reference's own trainer base, when its package is installed next to us
    from dinounet.training.nnUNetTrainer.nnUNetTrainerNoDeepSupervision import nnUNetTrainerNoDeepSupervision  # type: ignore
    from dinounet.api import evaluate, plan_and_preprocess, training  # type: ignore
    HAVE_NNUNET = True
except Exception:  # noqa: BLE001
    HAVE_NNUNET = False
    plan_and_preprocess = training = evaluate = None

    class nnUNetTrainerNoDeepSupervision:  # stand-in: only the class-level plug-in hook is used without nnU-Net
        enable_deep_supervision = False


class DinoUNetTrainer(nnUNetTrainerNoDeepSupervision):
    """DT:833-881."""
    _network_config = None
    _dinov3_pretrained_path = None
    _dinov3_model_name = None

    @classmethod
    def set_network_config(cls, network_config, dinov3_pretrained_path=None, dinov3_model_name=None, adapter_type="default"):
        cls._network_config = network_config
        if dinov3_pretrained_path is not None:
            cls._dinov3_pretrained_path = dinov3_pretrained_path
        if dinov3_model_name is not None:
            cls._dinov3_model_name = dinov3_model_name
        DinoUNetTrainer._network_config = cls._network_config                 # DT:853-855
        DinoUNetTrainer._dinov3_model_name = cls._dinov3_model_name
        DinoUNetTrainer._dinov3_pretrained_path = cls._dinov3_pretrained_path

    @staticmethod
    def build_network_architecture(architecture_class_name: str, arch_init_kwargs: dict,
                                   arch_init_kwargs_req_import: Union[List[str], Tuple[str, ...]], num_input_channels: int,
                                   num_output_channels: int, enable_deep_supervision: bool = True) -> nn.Module:
        config = DinoUNetTrainer._network_config.copy()
        config["architecture"] = config["architecture"].copy()
        config["architecture"]["deep_supervision"] = enable_deep_supervision
        return DinoUNet.from_config(network_config=config, input_channels=num_input_channels, num_classes=num_output_channels,
                                    dinov3_pretrained_path=DinoUNetTrainer._dinov3_pretrained_path,
                                    dinov3_model_name=DinoUNetTrainer._dinov3_model_name)


class DinoUNetTrainer_s(DinoUNetTrainer):
    _dinov3_model_name = "dinounet_s"
    _dinov3_pretrained_path = "dinounet/checkpoints/dinov3_vits16_pretrain_lvd1689m-08c60483.pth"


class DinoUNetTrainer_b(DinoUNetTrainer):
    _dinov3_model_name = "dinounet_b"
    _dinov3_pretrained_path = "dinounet/checkpoints/dinov3_vitb16_pretrain_lvd1689m-73cec8be.pth"


class DinoUNetTrainer_l(DinoUNetTrainer):
    _dinov3_model_name = "dinounet_l"
    _dinov3_pretrained_path = "dinounet/checkpoints/dinov3_vitl16_pretrain_lvd1689m-8aa4cbdd.pth"


class DinoUNetTrainer_7b(DinoUNetTrainer):
    _dinov3_model_name = "dinounet_7b"
    _dinov3_pretrained_path = "dinounet/checkpoints/dinov3_vit7b16_pretrain_lvd1689m-a955f4ea.pth"


DINOV3_TRAINERS = {"dinounet_s": DinoUNetTrainer_s, "dinounet_b": DinoUNetTrainer_b, "dinounet_l": DinoUNetTrainer_l,
                   "dinounet_7b": DinoUNetTrainer_7b}


def get_dinov3_trainer(model_name: str):
    if model_name not in DINOV3_TRAINERS:
        raise ValueError(f"Unsupported model: {model_name}. Supported models: {list(DINOV3_TRAINERS.keys())}")
    return DINOV3_TRAINERS[model_name]


def main_dinov3(model_name: str = "dinounet_s", dataset_id: int = 4, num_epochs: int = 50):
    """DT:958-1050: plan/preprocess -> set_network_config -> training -> evaluate, through the reference's nnU-Net façade."""
    if not HAVE_NNUNET:
        raise ImportError("main_dinov3 needs the reference's nnU-Net fork (package `dinounet` with batchgenerators etc.); "
                          "this repository replaces only the network hot path behind DinoUNetTrainer.build_network_architecture")
    trainer_class = get_dinov3_trainer(model_name)
    plans_identifier, network_configs = plan_and_preprocess(dataset_id=dataset_id, verify_dataset_integrity=True,
                                                            force_target_shape=[512, 512], force_n_stages=4,
                                                            configurations=["2d"], verbose=True, force_rerun=False)
    config = network_configs["2d"]
    trainer_class.set_network_config(config)
    result_folder, training_log = training(dataset_id=dataset_id, configuration="2d", trainer_class=trainer_class,
                                           plans_identifier=plans_identifier, initial_lr=0.001, num_epochs=num_epochs,
                                           batch_size=config["data_config"]["batch_size"])
    results = evaluate(dataset_id=dataset_id, result_folder=result_folder)
    return result_folder, training_log, results
