import os, sys, json
os.environ.setdefault("DINOUNET_ALLOW_RANDOM_BACKBONE","1")
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from dinounet_amd.plans import PLANS_2D
from dinounet_amd import ops
from dinounet_amd.network_architecture import DinoUNet
from dinounet_amd.training import TrainStep
from dinounet_amd.optim import FusedClipSGD
dev = torch.device("cuda", 0)
torch.manual_seed(1234)
net = DinoUNet.from_config(PLANS_2D, 3, 2, dinov3_pretrained_path=None, dinov3_model_name="dinounet_l", precision="bf16").to(dev).train()
params = [p for p in net.parameters() if p.requires_grad]
opt = FusedClipSGD(params, lr=1e-3, momentum=0.99, nesterov=True, weight_decay=3e-5, max_norm=12.0)
x = torch.randn(8, 3, 512, 512, device=dev); tgt = torch.randint(0, 2, (8, 1, 512, 512), device=dev)
ts = TrainStep(net, opt, params, x.shape, tgt.shape, dev, graph=False)
ts(x, tgt); ts(); torch.cuda.synchronize()
ops.PROFILE = ops.KernelProfile(detail=True)
for _ in range(2): ts()
torch.cuda.synchronize()
roof, brk = ops.PROFILE.roofline(2500.0, 8000.0, 2)
agg = {}
for name, e0, e1, fl, nb in ops.PROFILE.rec:
    a = agg.setdefault(name, [0.0, 0]); a[0] += e0.elapsed_time(e1) * 1e3 / 2; a[1] += 0.5
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][0])[:int(os.environ.get("DETAIL_TOP", "70"))]:
    print(f"{v[0]:9.1f} us/step  x{v[1]:4.1f}  {k}")
