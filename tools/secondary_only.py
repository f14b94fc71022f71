"""The secondary workloads of bench.py alone (TSP-500 batch 16 pomo 500; VRPLIB X-n1001 x8), for rocprofv3 runs."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
torch.cuda.set_device(0)
import bench
print(json.dumps(bench.secondary_workloads("cuda:0")))
