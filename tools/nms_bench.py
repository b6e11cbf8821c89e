import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import yolo_for_turbines_amd as yt
from tests import golden_inputs as gi
images, n, nc = 16, 10000, int(sys.argv[1]) if len(sys.argv) > 1 else 80
kind = sys.argv[2] if len(sys.argv) > 2 else "uniform"
gen = gi.boxes_uniform if kind == "uniform" else (lambda n, nc, s: gi.boxes_clustered(n, nc, s, jitter=0.15))
t = torch.from_numpy(np.stack([gen(n, nc, 1000 + b) for b in range(images)])).cuda()
for _ in range(3):
    keep, count = yt.nms_indices(t, 0.45, 0.5, "center")
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    keep, count = yt.nms_indices(t, 0.45, 0.5, "center")
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 10
print(f"{kind} nc={nc}: {dt*1e3:.3f} ms per {images}x{n} -> {images*n/dt/1e6:.1f} M boxes/s, kept mean {float(count.float().mean()):.0f}")
