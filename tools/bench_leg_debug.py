import sys, os, json
import torch  # the bench process has PyTorch (and its own ROCm libraries) loaded
sys.path.insert(0, os.getcwd())
import bench
from akari_render_amd import capi
ctx = capi.Context(0)
print(json.dumps(bench.textured_room_leg(ctx), indent=0)[:3000])
