import sys, json
sys.path.insert(0, "/root/repo")
import torch, bench
import yolo_for_turbines_amd as yt
print(json.dumps(bench.decode_bench(yt, torch.device("cuda:0"))))
print(json.dumps(bench.decode_bench(yt, torch.device("cuda:0"), batch=128, size=608)))
