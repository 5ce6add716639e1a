import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import runpy, torch
from torch.profiler import profile, ProfilerActivity
sys.argv = ["sdp_c4_probe.py", "1024", "1e-4"]
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    runpy.run_path(os.path.join(os.path.dirname(__file__), "sdp_c4_probe.py"), run_name="__main__")
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=22, max_name_column_width=70))
