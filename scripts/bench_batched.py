"""bench.py's batched-windows line alone (default B = 8, 32).  Run on the GPU box."""
import sys, json
sys.path.insert(0, '.')
import bench, argparse
ap = argparse.ArgumentParser(); ap.add_argument("--steps", type=int, default=100); ap.add_argument("--warmup", type=int, default=10); ap.add_argument("--no-prior", action="store_true")
ap.add_argument("--B", type=int, nargs="*", default=[8, 32]); ap.add_argument("--min-timed-s", type=float, default=0.2)
args = ap.parse_args()
print(json.dumps(bench.batched_line(args, 0, Bs=tuple(args.B), min_timed_s=args.min_timed_s)))
