"""bench.py's multi-rank control flow, rehearsed on the one-GPU box: two ranks share the GPU, torch.distributed runs on
gloo (device tensors staged through the host) -- the branch a driver takes on an 8-GPU node (shard seeds, statistics
all-reduce, dcv_mlp_dp_step with its all-reduce callback, MAX of the elapsed times, rank-0 JSON line) has then executed
once before it meets RCCL."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_bench_world2_control_flow():
    port = _free_port()
    args = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--frames", "400000", "--batch", "4096",
            "--steps", "6", "--warmup", "2", "--no-cpu-baseline", "--large-batch", "0", "--other-mode-steps", "0", "--profile-every", "2"]
    procs = []
    for rank in (0, 1):   # fresh child processes: nothing here has touched the GPU on their behalf
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen(args, env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=600) for p in procs]
    for p, (so, se) in zip(procs, outs):
        assert p.returncode == 0, se[-2000:]
    js = [[l for l in so.splitlines() if l.startswith("{")] for so, _ in outs]   # gloo prints its own connection chatter on stdout
    assert len(js[0]) == 1 and js[1] == []                # only rank 0 prints the JSON line
    line = json.loads(js[0][0])
    assert line["n_gpus"] == 2 and line["steps"] == 6 and line["scaling"] == "strong"
    assert line["metric"].startswith("Deep-TICA training frames/sec")
    assert line["config"]["parallelism"] == "frame-shard dp2" and line["config"]["global_batch"] == 4096
    assert line["value"] > 0 and abs(line["value"] - 6 * 4096 / (line["ms_per_step"] * 6e-3)) < 1e-6 * line["value"]
    assert line["roofline"] is not None and line["roofline"]["rows_per_launch"] == 2048 + 10
    assert line["loss_first"] is not None and -4.0 <= line["loss_last_train"] <= 0.0   # -sum(eig^2) of a d = 4 TICA
    assert "cpu_baseline" not in line


def test_bench_self_launches_its_ranks():
    """`python bench.py --gpus 2` with NO launcher environment: the script itself starts one fresh child
    process per rank before it touches the GPU, relays rank 0's JSON line and exits 0.  Two ranks share the one GPU over
    gloo; the line carries the communicator's world size, the measured all-reduce times, the large-batch and shuffled
    blocks of the N > 1 path."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    args = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--frames", "600000", "--batch", "4096",
            "--steps", "6", "--warmup", "2", "--no-cpu-baseline", "--large-batch", "65516", "--large-steps", "3", "--other-mode-steps", "0",
            "--shuffled-steps", "5"]
    p = subprocess.run(args, env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["steps"] == 6 and line["scaling"] == "strong"
    cfg = line["config"]
    assert cfg["parallelism"] == "frame-shard dp2" and cfg["communicator_world_size"] == 2
    ct = cfg["collective_timing"]
    assert ct["statistics"]["bytes"] == 8 * (2 * 4 + 2 * 16) and ct["statistics"]["us_per_allreduce"] > 0
    assert ct["gradients"]["bytes"] == 4 * cfg["params"] and ct["gradients"]["us_per_allreduce"] > 0
    # the two layer-0 products are sampled on different steps (every 4th each), never on the first one behind the barrier
    assert line["roofline"]["samples"] == {"layer0.fwd": 2, "layer0.wgrad": 1} and line["roofline"]["rows_per_launch"] == 2048 + 10
    assert line["large_batch"]["global_batch"] == 65516 and line["large_batch"]["value"] > 0
    assert line["shuffled"]["rows_per_launch"] == 2 * 2048 and line["shuffled"]["value"] > 0
    assert "c2" not in line and "ref_small" not in line      # single-GPU blocks


def test_bench_refuses_more_rccl_ranks_than_gpus():
    """RCCL wants one device per rank: with fewer GPUs than ranks the self-launcher says so instead of hanging in init."""
    import torch

    if torch.cuda.device_count() >= 2:
        pytest.skip("needs a box with fewer GPUs than ranks")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], env=env, cwd=ROOT, stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, text=True, timeout=300)
    assert p.returncode != 0 and "one device per rank" in p.stderr


def test_bench_single_gpu_line_carries_every_block():
    """The driver's form (`python bench.py --gpus 1 --steps K --warmup W`) on a small matrix: ONE JSON line with the contract
    keys, `roofline` and `cpu_baseline`, and the round-4 blocks (`shuffled`, `large_batch`, `c2`, `ref_small`) -- with the
    epoch loop issued as dcv_mlp_train_steps runs between the sampled steps and the validation passes as dcv_mlp_eval_steps."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    args = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--frames", "300000", "--batch", "4096", "--steps", "70", "--warmup", "3",
            "--cpu-seconds", "1", "--large-batch", "65516", "--large-steps", "3", "--other-mode-steps", "4", "--shuffled-steps", "40",
            "--c2-steps", "420", "--ref-small-steps", "12", "--fit-epochs", "2"]
    p = subprocess.run(args, env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    line = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
                "config", "roofline", "cpu_baseline"):
        assert key in line, key
    assert line["n_gpus"] == 1 and line["steps"] == 70 and line["warmup"] == 3 and line["vs_baseline"] is None and line["dtype"] == "f32"
    assert abs(line["value"] - 4096 / (line["ms_per_step"] * 1e-3)) < 1e-6 * line["value"]
    # 300 000 frames, lag 10, 80 % training pairs: 58 steps per epoch -> one validation pass inside the 70 timed steps
    assert line["config"]["validation_steps_timed"] >= 1
    r = line["roofline"]
    assert r["bound"] == "mfma" and 0 < r["frac"] < 1 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9 and r["rows_per_launch"] == 4096 + 10
    cb = line["cpu_baseline"]
    assert cb["kind"] == "port" and cb["value"] > 0 and cb["cores"] >= 1 and line["value"] > 10 * cb["value"]
    assert line["shuffled"]["value"] > 0 and line["shuffled"]["rows_per_launch"] == 2 * 4096
    assert line["large_batch"]["global_batch"] == 65516 and line["large_batch"]["value"] > 0
    c2 = line["c2"]
    assert c2["value"] > 0 and c2["config"]["validation_steps_timed"] == 2 * c2["config"]["val_steps_per_epoch"] > 0   # 420 steps = two epochs of 195
    assert c2["roofline"]["samples"] > 0
    rs = line["ref_small"]["runs"]
    assert len(rs) == 4 and all(x["fused_small_network_path"] and x["value"] > 0 for x in rs)
    fits = line["calculator_fit"]["runs"]
    assert [(x["cv"], x["batch"]) for x in fits] == [("deep_tica", 256), ("deep_tica", 4096), ("ae", 256), ("ae", 4096)]
    assert all(x["ok"] and x["epochs"] == 2 and x["seconds"] > 0 for x in fits)


@pytest.mark.parametrize("native", [False, True])
def test_bench_collective_path_on_real_rccl_world1(native):
    """DCV_FORCE_DIST=1: the data-parallel branch of bench.py (dcv_mlp_dp_step, statistics and gradient all-reduces in stream
    order with the kernels) over the REAL RCCL -- torch.distributed's `nccl` backend, or the library's own communicator with
    --native-rccl -- as a one-rank group on the one GPU.  The closest rehearsal of the N > 1 run this box allows: RCCL
    initialisation, its kernels between ours on the launch stream, the side-stream overlap of the upper layers' gradients."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["DCV_FORCE_DIST"] = "1"
    env["MASTER_PORT"] = str(_free_port())
    args = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--frames", "300000", "--batch", "4096", "--steps", "12", "--warmup", "3",
            "--no-cpu-baseline", "--large-batch", "0", "--other-mode-steps", "0", "--shuffled-steps", "6", "--c2-steps", "0", "--ref-small-steps", "0", "--fit-epochs", "0"]
    if native:
        args.append("--native-rccl")
    p = subprocess.run(args, env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    line = json.loads(lines[0])
    assert line["n_gpus"] == 1 and line["value"] > 0 and line["config"]["communicator_world_size"] == 1
    ct = line["config"]["collective_timing"]
    assert ct["statistics"]["us_per_allreduce"] > 0 and ct["gradients"]["us_per_allreduce"] > 0
    assert -4.0 <= line["loss_last_train"] <= 0.0
