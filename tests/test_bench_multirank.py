"""CPU, world_size 2, gloo: the REAL bench.py N > 1 branch (process-group set-up from the torchrun environment, barriers,
MAX / SUM reductions, the single result gather, rank-0 JSON line) with a scripted utterance runner (`--stub`): what the
driver launches at N = 2, 4, 8, minus the kernels."""
import json
import os
import socket
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run(world, steps=2):
    env = dict(os.environ, FQ3_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--steps", str(steps),
           "--warmup", "1", "--stub", "--config3-utterances", "10"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout          # rank 0 prints ONE JSON line
    return json.loads(lines[0])


def test_bench_two_ranks_gloo_stub():
    d = _run(2)
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["warmup"] == 1 and d["scaling"] == "weak"
    assert d["higher_is_better"] is True and d["unit"] == "x real-time"
    # whole-job aggregate: 2 ranks x 2 utterances x 200 frames of 80 ms over the MAX wall time
    audio_s = 2 * 2 * 200 * 0.08
    assert abs(d["value"] * d["ms_per_step"] * d["steps"] / 1000 - audio_s) < 0.05 * audio_s
    assert d["gathered_samples"] >= 2 * 1000          # both ranks' PCM arrived through the gather
    c3 = d["config3_sharded_batched"]
    assert c3["utterances"] == 10                     # 5 + 5 lengths came back through the same gather
    assert "stub" in d["data"]
    # the line proves its own rank count (an all-reduce of ones through the collective backend) and carries every rank's own figures
    assert d["rccl_ranks"] == 2 and d["collective_backend"] == "gloo"
    pr = d["per_rank"]
    assert len(pr["value"]) == 2 and len(pr["ttfa_ms_p50"]) == 2 and all(v > 0 for v in pr["value"]) and all(t > 0 for t in pr["ttfa_ms_p50"])
    # whole-job value = all ranks' audio over the MAX wall: never above the sum of the ranks' own rates
    assert d["value"] <= sum(pr["value"]) * 1.001


def test_bench_single_rank_stub_has_same_shape():
    d = _run(1, steps=1)
    assert d["n_gpus"] == 1 and d["config3_sharded_batched"]["utterances"] == 10
    assert d["rccl_ranks"] == 1 and len(d["per_rank"]["value"]) == 1 and abs(d["per_rank"]["value"][0] - d["value"]) <= 0.02 * d["value"]


def test_bench_gpus_flag_alone_launches_the_ranks():
    """`python bench.py --gpus 2 ...` WITHOUT a launcher (the form README / DESIGN advertise and the driver's SCALE command
    uses at N = 1): bench.py starts the two ranks itself and rank 0's line says n_gpus: 2."""
    env = dict(os.environ, FQ3_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--stub",
           "--config3-utterances", "6"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config3_sharded_batched"]["utterances"] == 6


def test_bench_refuses_a_world_that_contradicts_gpus():
    """Launched as 2 ranks but told --gpus 1: every rank exits non-zero instead of printing a line with the wrong n_gpus."""
    env = dict(os.environ, FQ3_BENCH_BACKEND="gloo")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "1", "--warmup", "0", "--stub"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]
