import sys, os, json
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, bench
dev = torch.device("cuda", 0)
frames = bench.make_inputs(dev, [0], 20480)
r = bench.train_step_summary(dev, frames[0])
print(json.dumps({k: r[k] for k in ("ms_per_step", "graphed", "breakdown")}, indent=1))
