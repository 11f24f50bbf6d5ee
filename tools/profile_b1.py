"""One clip, one lane, replayed from its hipGraph 200 times (bench.py's gpu_b1 leg without the timing): the command to put behind
`rocprofv3 --kernel-trace --stats --output-format csv -d <dir> -- python3 tools/profile_b1.py` for the per-kernel picture of configs[0] on the GPU."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from emotiongestures_amd.pipeline import ClipPipeline

dev = torch.device("cuda:0")
gen, vae, mel, _, _ = bench.build_models("bf16x3", dev)
inp = bench.make_inputs(1, 1000)
g1 = {k: torch.from_numpy(v[:1]).to(dev) for k, v in inp.items()}
pipe = ClipPipeline((gen, vae, mel), g1, dev, lanes=1, branch_streams=(os.environ.get("EG_B1_BRANCH", "1") == "1"))
for _ in range(int(os.environ.get("EG_B1_REPS", "200"))):
    pipe.wait(pipe.launch_next())
torch.cuda.synchronize()
print("done")
