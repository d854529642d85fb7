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


GEMM_BIN = os.path.join(ROOT, "tools", "microbench", "gemm_bench")


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(200, 1024, 3072), (52, 1536, 2048), (1000, 768, 1536), (2000, 1024, 2048)])
def test_every_gemm_kernel_family_agrees_bit_for_bit(shape):
    """Round 6: the implicit-GEMM families (register-prefetch tiles, the LDS-DMA tile by four / eight waves in 128- and 64-row forms and at
    ring depths 2 .. 6, the chain GEMM, gemm_launch's own choice) under the plain and the XCD-aware tile order compute every output
    element by the same ascending chain of 32-wide MFMA products: on random operands their outputs are identical to the two-stage
    four-wave LDS-DMA tile's, element for element -- ragged last tiles in M and N included (200 and 1000 rows; 52 rows = one ragged tile).
    A tail decode of the codec relies on it (a chunk's GEMMs and the full decode's take different kernels)."""
    assert os.path.exists(GEMM_BIN), "tools/microbench/gemm_bench missing: run __graft_entry__.build() (make tools)"
    out = subprocess.run([GEMM_BIN, "2", "glds"] + [str(v) for v in shape], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    lines = [l for l in out.stdout.splitlines() if l.startswith("M=")]
    product = [l for l in lines if "dbg " not in l]                 # (the measurement builds without MFMAs / fragment reads differ by design)
    assert len(product) >= 20, out.stdout
    assert sum("(reference)" in l for l in product) == 1
    assert all(("bit-identical" in l) or ("(reference)" in l) for l in product), "\n".join(l for l in product if "MISMATCH" in l)
