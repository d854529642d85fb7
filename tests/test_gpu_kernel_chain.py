"""GPU: the kernel-tuning harness (tools/microbench/kernel_chain, built by __graft_entry__.build()) launches the
PRODUCT decode kernels standalone -- no torch, no Python in the loop -- and checks them against double-precision
CPU references: DPP / permlane-swap lane reductions, the RMSNorm / SwiGLU / residual GEMV variants at the 0.6B
shapes, the one-wave-per-head predictor attention (output + KV append) and the split-KV attention + merge path."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "tools", "microbench", "kernel_chain")


@pytest.mark.gpu
def test_product_kernels_standalone_self_check():
    assert os.path.exists(BIN), "tools/microbench/kernel_chain missing: run __graft_entry__.build() (make tools)"
    out = subprocess.run([BIN, "check", "2"], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    lines = [l for l in out.stdout.splitlines() if l.startswith("check ")]
    assert len(lines) >= 8, out.stdout
    assert all(l.rstrip().endswith("ok") for l in lines), out.stdout
    assert "SELF-CHECK FAILURES" not in out.stdout
