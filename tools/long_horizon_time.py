"""bench.py's secondary.long_horizon leg on its own (unicycle, N = 512, batch 1024: per-pass launches, factor workspace in HBM) -- for a rocprofv3 kernel trace:
    rocprofv3 --kernel-trace --stats ... -- python tools/long_horizon_time.py        (tools/collect_profiles.sh: profiles/rNN_long_kernel_stats.csv)
"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench   # noqa: E402

print(json.dumps(bench.long_horizon_leg(0)))
