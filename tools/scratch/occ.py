import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from dinounet_amd import _lib
L = _lib.lib()
torch.zeros(1, device="cuda")
for w in (0, 1, 2):
    print("occupancy query", w, L.du_debug_attn_occupancy(w))
