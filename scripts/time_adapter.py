"""bench.py's `adapter` line alone: wall time of ldso::GpuBackend::optimize(6) against the reference's FullSystem::optimize(6) on the same reference
object graph, with the adapter's time split.  Run on the GPU box."""
import json, sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench
print(json.dumps(bench.adapter_line(sys.argv[1] if len(sys.argv) > 1 else "C3")))
