import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import torch
from test_guard_resume_gpu import _module, _trainable
mod, model, batches = _module()
for i in range(3):
    mod.training_step({"image": batches[i][0], "target": batches[i][1]}, i)
eng = model._engine
ckpt = mod.checkpoint_state()
p3 = _trainable(model)
m3, v3 = eng._flat.m.clone(), eng._flat.v.clone()
model.cuda(); model.load_state_dict(model.state_dict())
mod.training_step({"image": batches[3][0], "target": batches[3][1]}, 3)
ref4 = _trainable(model)
print("a: m/v restored equal before step? step", eng._flat.step)
mod2, model2, _ = _module(seed=5)
mod2.load_checkpoint_state(ckpt)
l3 = _trainable(model2)
print("after load max diff", max(float((l3[k] - p3[k]).abs().max()) for k in p3))
sd1, sd2 = model.state_dict(), model2.state_dict()
e2 = model2._engine
print("stash", e2._opt_stash is not None, e2._flat is None)
mod2.training_step({"image": batches[3][0], "target": batches[3][1]}, 3)
print("m diff after", float((e2._flat.m - eng._flat.m).abs().max()), float((e2._flat.v - eng._flat.v).abs().max()), e2._flat.step)
got4 = _trainable(model2)
for k in ref4:
    d = float((got4[k] - ref4[k]).abs().max())
    if d > 1e-5:
        print(k, d, float((ref4[k] - p3[k]).abs().max()), float((got4[k] - p3[k]).abs().max()))
