import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import torch, numpy as np
from oracle import vit_oracle as vo
from oracle.closed_form import *
from gpu_util import native_model
cfg = vo.VitConfig(img_size=48, embed_dim=128, depth=2, num_heads=2, init_values=0.1)
model, sd = native_model(cfg); model.eval()
B=3
x = closed_form_images("t48/0", B, 48); mask = exact_masks(B, 9, 4, 1)
ref_all = vo.forward(sd, cfg, x, mask, True)
ref_feat = vo.forward_features(sd, cfg, x, mask, None)   # normed
got = model(x.cuda(), mask.cuda(), return_all_tokens=True).cpu()
err = (got-ref_all).abs()
print("per-row max err:\n", err.amax(-1))
print("ref row absmax:\n", ref_all.abs().amax(-1))
print("mask:\n", mask.view(B,-1))
ends = vo.forward(sd, cfg, x, mask, True, 'end')
e = model.engine(B)
xl = e.ws_tensor("x", 2, (B,10,128)).cpu()
# note: last forward used mask -> compare to student residual
xs = vo.forward_features(sd, cfg, x, mask, 'end')[-1]
print("resid err", (xl-xs).abs().max().item(), "resid std per row", xs.std(-1))
