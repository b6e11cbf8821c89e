import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from oracle import net as onet
from tests import golden_inputs as gi
import yolo_for_turbines_amd as yt
from yolo_for_turbines_amd import engine
from tests.test_gpu_parity import _block
for i in [9, 10, 15, 16, 22, 3]:
    cin, cout, k, s, bn, h = gi.BLOCK_CONFIGS[i]
    for tile in [1,2,3,4,5,6]:
        blk, x = _block(yt, i, "leaky_relu")
        engine._module_state.tile_override = tile
        try:
            with torch.no_grad():
                y = blk(x.cuda()).cpu()
        except Exception as e:
            print(i, tile, "ERR", str(e)[:60]); engine._module_state._packed.clear(); continue
        p = gi.block_params(i, cin, cout, k, bn)
        sd = {"b.conv.weight": torch.from_numpy(p["w"])}
        if bn:
            sd.update({"b.batch_norm.weight": torch.from_numpy(p["gamma"]), "b.batch_norm.bias": torch.from_numpy(p["beta"]),
                       "b.batch_norm.running_mean": torch.from_numpy(p["mean"]), "b.batch_norm.running_var": torch.from_numpy(p["var"])})
        else:
            sd["b.conv.bias"] = torch.from_numpy(p["bias"])
        with torch.no_grad():
            ref = onet.cnn_block(sd, dict(prefix="b", cin=cin, cout=cout, k=k, stride=s, bn=bn), x, "leaky_relu")
        err = (y-ref).abs()
        bad = (err > 1e-4)
        print(i, (cin,cout,k,s,h), "tile", tile, "maxerr %.3g" % err.max().item(), "bad frac %.3f" % bad.float().mean().item(),
              "bad by cout-block:", [round(bad[:, c:c+64].float().mean().item(),2) for c in range(0, cout, 64)][:8], flush=True)
