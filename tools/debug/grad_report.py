import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from oracle import VIT_CONFIGS, det_state_dict, synth_batch, weighted_mse_loss
from oracle.model import generator_state_shapes, OracleTrainer, orion_marker_weights
from miphei_vit_amd.generators import get_vitmatte

cfgname, img, nc, B, seed = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), 11
cfg = VIT_CONFIGS[cfgname]
sd = det_state_dict(generator_state_shapes(cfg, img, nc), seed=seed, layerscale=0.5)
p = {k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}
model = get_vitmatte(cfgname, img, nc, use_lora=True, pretrained=False)
model.load_state_dict(p); model = model.cuda()
x, y = synth_batch(seed, B, img, nc)
w = orion_marker_weights(nc)
tr = OracleTrainer(p, cfg, nc, batch_size=B, total_iters=100, weights=w)
out_ref, loss_ref, gref = tr.loss_and_grads(x, y)
model.train()
out = model(x.cuda())
loss = weighted_mse_loss(y.cuda(), out, w.cuda())
loss.backward()
print("loss", float(loss), float(loss_ref))
named = dict(model.named_parameters())
rows = []
for k, gr in gref.items():
    got = named[k].grad.detach().cpu().double(); gr = gr.double()
    rel = float((got - gr).norm() / gr.norm().clamp_min(1e-30))
    rows.append((rel, k, float(gr.norm()), float(got.norm())))
for rel, k, a, b in sorted(rows, reverse=True):
    print(f"{rel:9.4f} {k:60s} ref|g|={a:.3e} got|g|={b:.3e}")
