import importlib, time, numpy as np, sys
sys.path.insert(0, '.')
pkg = importlib.import_module("mp-gadget_amd")
import torch
ics = pkg.ics
t = time.time(); a, _, box = ics.s_zel(320); ta = time.time() - t
ics.ZEL_TORCH_MIN = 10 ** 9
t = time.time(); b, _, _ = ics.s_zel(320); tb = time.time() - t
d = np.abs(a - b); d = np.minimum(d, box - d)
print("320^3: torch %.1f s numpy %.1f s  max |dpos| %.3e (spacing %.1f)  in (0,box]: %s" % (ta, tb, d.max(), box / 320, bool((a > 0).all() and (a <= box).all())))
ics.ZEL_TORCH_MIN = 320
t = time.time(); c, _, box = ics.s_zel(512); print("512^3 torch: %.1f s" % (time.time() - t), c.shape, torch.cuda.memory_allocated() >> 20, "MiB still allocated")
