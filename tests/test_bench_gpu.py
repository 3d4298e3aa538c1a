"""bench.py end to end on the GPU box: the one-GPU line (with its parity check against the oracle) and the N > 1 code path
(two ranks on ONE GPU over gloo, LDSO_BENCH_ONE_GPU=1 - RCCL refuses two ranks on one device) so that the sharded iteration
(set_shard -> gn_reduce_local -> all_reduce -> gn_solve_reduced) is executed every round, not only on an 8-GPU node."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _last_json(out):
    lines = [l for l in out.splitlines() if l.startswith("{")]
    assert lines, out[-2000:]
    return json.loads(lines[-1])


def test_bench_single_gpu_line_with_parity():
    r = subprocess.run([sys.executable, "bench.py", "--steps", "20", "--warmup", "5", "--no-cpu-baseline", "--no-extras"], cwd=ROOT,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    j = _last_json(r.stdout)
    assert j["n_gpus"] == 1 and j["steps"] == 20 and j["value"] > 0 and j["state_finite"]
    assert j["timed_total_ms"] >= 50.0 and j["timed_blocks"] >= 3
    assert j["parity_vs_oracle"]["ok"] and j["parity_vs_oracle"]["rel"] <= 1e-4
    rf = j["roofline"]
    assert rf["bound"] == "hbm" and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-4
    assert abs(rf["achieved"] - rf["algorithmic_bytes_per_launch"] / (rf["avg_launch_us"] * 1e-6) / 1e9) < 0.01 * rf["achieved"]      # live figure


def test_bench_two_ranks_on_one_gpu():
    env = dict(os.environ, LDSO_BENCH_ONE_GPU="1", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29517", "bench.py", "--gpus", "2", "--steps", "5", "--warmup", "2", "--no-cpu-baseline"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    j = _last_json(r.stdout)
    assert j["n_gpus"] == 2 and j["value"] > 0 and j["state_finite"]
    assert "sharded over 2 GPUs" in j["config"]["parallelism"]
    pv = j["parity_vs_oracle"]          # rank 0 gathered the shards, the oracle re-evaluated the state and the 10-iteration log
    assert pv["ok"] and pv["ranks"] == 2 and pv["rel"] <= 1e-4 and pv["energy_log_10_iterations_max_rel"] <= 1e-4 and pv["frame_states_max_abs_diff_between_ranks"] == 0.0
    assert j["rccl_ranks"] == 0          # gloo debug mode


def test_bench_two_ranks_on_one_gpu_with_the_peer_write_exchange():
    """the same two ranks with --allreduce p2p: the receive windows cross the process boundary as hipIpcMemHandle_t (all_gather_object over the
    process group = control plane only), the iteration is ldso_ba_enqueue_gn_p2p"""
    env = dict(os.environ, LDSO_BENCH_ONE_GPU="1", MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29519", "bench.py", "--gpus", "2", "--steps", "5", "--warmup", "2", "--no-cpu-baseline", "--allreduce", "p2p", "--min-timed-s", "0.2"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    j = _last_json(r.stdout)
    assert j["n_gpus"] == 2 and j["value"] > 0 and j["state_finite"]
    assert "peer-write" in j["config"]["parallelism"]
    assert j["parity_vs_oracle"]["ok"] and j["parity_vs_oracle"]["ranks"] == 2
