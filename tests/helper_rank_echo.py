"""Stand-in rank program for tests/test_bench_launcher.py (CPU): rendezvous over gloo as bench.py's ranks do, all-reduce the ranks,
rank 0 prints one JSON line; `--fail` makes rank 1 exit non-zero (the launcher must relay that)."""
import json
import os
import sys

import torch
import torch.distributed as dist

if __name__ == "__main__":
    dist.init_process_group("gloo")
    t = torch.tensor([float(dist.get_rank() + 1)])
    dist.all_reduce(t)
    if "--fail" in sys.argv and dist.get_rank() == 1:
        sys.exit(3)
    if dist.get_rank() == 0:
        print(json.dumps({"world_size": dist.get_world_size(), "sum": float(t.item()), "args": sys.argv[1:],
                          "ipc_legacy": os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY")}), flush=True)
    dist.destroy_process_group()
