mkdir -p gpurun_out/r03d
python tools/level_stats.py 1024 prospero.vm > gpurun_out/r03d/level_stats.txt 2>&1
FHIP_NO_COLUMN_INV=1 python tools/level_stats.py 1024 prospero.vm > gpurun_out/r03d/level_stats_general.txt 2>&1
python tools/wave_stats.py 1024 prospero.vm > gpurun_out/r03d/wave_stats.txt 2>&1
python - <<'P' > gpurun_out/r03d/tile_v.txt 2>&1
import os, sys
os.environ["FHIP_PROBE"]="1"; os.environ["FHIP_NO_PIPELINE"]="1"
sys.path.insert(0, os.getcwd())
import torch, fidget_amd as F
hip = F.HipContext(0, torch.cuda.current_stream().cuda_stream)
s = F.Shape.from_vm("models/prospero.vm", hip=hip)
out = torch.zeros((1024,1024,4), dtype=torch.int32, device="cuda")
for inv in (0, 1):
    hip.set_option("no_column_inv", inv)
    for _ in range(2): F.render3d(s, 1024, out=out)
    hip.sync(); hip.wave_stats()
    print("no_column_inv", inv, hip.tile_v)
P
cat gpurun_out/r03d/tile_v.txt
