import sys, json
sys.path.insert(0, '.')
import bench, argparse
args = argparse.Namespace(steps=100, warmup=10, no_prior=False)
print(json.dumps(bench.batched_line(args, 0, Bs=(1, 2))))
