import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import bench, torch
from mnk import configs, workload
cfg = configs.get("moving-gif"); dev = torch.device("cuda:0")
src, drv = workload.synthetic_pair(32, 64, 64)
x = {"source": src.to(dev), "video": drv.to(dev)}
r = bench.dropin_loop(cfg, x, dev, 8, 3, mnk_adam=bool(int(sys.argv[1])))
print(r["ms_per_step"])
