"""MI355X, one GPU: the RCCL leg of the N > 1 path executes for real -- process group with backend "nccl" (= RCCL on ROCm),
world size 1, the result gather (`gather_arrays`: lengths all_gather + padded payload all_gather) and the MAX / SUM all-reduces
bench.py issues, all on device tensors.  (Two or more GPUs are the driver's to launch; the N = 2 control flow runs under gloo in
tests/test_bench_multirank.py and tests/test_sharding_gloo.py.)"""
import os
import subprocess
import sys
import textwrap

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu

_SCRIPT = textwrap.dedent("""
    import os, sys
    sys.path.insert(0, os.path.join(%r, "faster-qwen3-tts_amd"))
    import numpy as np, torch, torch.distributed as dist
    from fq3hip.sharding import gather_arrays, run_sharded
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29617")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    assert dist.get_backend() == "nccl"
    a = np.arange(1000, dtype=np.float32) * 0.5
    parts = gather_arrays(a, "cuda:0", always_collective=True)          # two RCCL all_gathers on device buffers
    assert len(parts) == 1 and np.array_equal(parts[0], a)
    t = torch.tensor([1.5, 2.5], device="cuda:0", dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX); dist.all_reduce(t)
    dist.barrier()
    torch.cuda.synchronize()
    assert t.tolist() == [1.5, 2.5]
    out = run_sharded([3, 4, 5], lambda v: np.full(v, float(v), np.float32), "cuda:0")
    assert [len(x) for x in out] == [3, 4, 5]
    dist.destroy_process_group()
    print("RCCL_OK")
""") % ROOT


def test_rccl_world1_gather_and_reductions_on_device():
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-c", _SCRIPT], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "RCCL_OK" in r.stdout, (r.stdout[-1000:], r.stderr[-3000:])
