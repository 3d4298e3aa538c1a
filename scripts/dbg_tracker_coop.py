import sys, os, json
sys.path.insert(0, "/root/repo")
import bench
os.environ.pop("LDSO_TR_NO_COOP", None)
r = bench.tracker_line(); print("COOP", json.dumps(r))
os.environ["LDSO_TR_NO_COOP"] = "1"
r = bench.tracker_line(); print("SOLO", json.dumps(r))
