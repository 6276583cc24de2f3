import sys, json
sys.path.insert(0, "tools")
import torch
from bench_configs import blockmix_case
r = blockmix_case("C3 fp32", 32, 256, 16, 72, 16, torch.float32, (4, 4), graph=True)
print(json.dumps({k: r[k] for k in ("ms", "hbm_frac", "ms_graph_replay", "hbm_frac_graph_replay", "dominant_kernel", "dominant_kernel_us", "kernel_us_per_step")}))
r = blockmix_case("C3 bf16", 32, 256, 16, 72, 16, torch.bfloat16, (4, 4), graph=True)
print(json.dumps({k: r[k] for k in ("ms", "hbm_frac", "ms_graph_replay", "hbm_frac_graph_replay", "dominant_kernel", "dominant_kernel_us", "kernel_us_per_step")}))
