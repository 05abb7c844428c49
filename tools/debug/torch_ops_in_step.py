"""Which torch (non-library) kernels still run inside one training step: torch.profiler table of aten ops by device time."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, runpy
sys.argv = ["bench.py", "--steps", "3", "--warmup", "2", "--no-cpu-baseline"]
from torch.profiler import profile, ProfilerActivity
import bench
orig = bench.main if hasattr(bench, "main") else None
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    orig()
print(prof.key_averages(group_by_input_shape=True).table(sort_by="cuda_time_total", row_limit=45, max_name_column_width=40, max_shapes_column_width=60))
