import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, numpy as np
import fidget_amd as F
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
free0 = torch.cuda.mem_get_info()[0]
hip = F.HipContext(0, torch.cuda.current_stream().cuda_stream)
hip.set_option("frame_lanes", 0)
p = F.Shape.from_vm(os.path.join(ROOT, "models", "prospero.vm"), hip=hip)
out = torch.zeros((1024, 1024, 4), dtype=torch.int32, device="cuda")
print("ctx", (free0 - torch.cuda.mem_get_info()[0]) / 2**20)
for no_inv in (0, 1):
    hip.set_option("no_column_inv", no_inv)
    for i in range(12):
        F.render3d(p, 1024, out=out)
        if i in (0, 3, 11):
            hip.sync(); print("no_inv", no_inv, "frame", i, "MB", (free0 - torch.cuda.mem_get_info()[0]) / 2**20, hip.counters())
out2 = torch.zeros((2048, 2048, 4), dtype=torch.int32, device="cuda")
for i in range(6):
    F.render3d(p, 2048, out=out2)
hip.sync(); print("2048 general MB", (free0 - torch.cuda.mem_get_info()[0]) / 2**20, hip.counters())
