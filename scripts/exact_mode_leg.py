import sys, json, argparse
sys.path.insert(0, "/root/repo")
import torch, bench
a = argparse.Namespace(height=480, width=640)
torch.cuda.set_device(0)
print(json.dumps(bench.exact_mode_leg(a, 0), indent=0))
