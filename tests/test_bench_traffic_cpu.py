"""CPU: the per-frame traffic accounting of bench.py (roofline.traffic): only dispatches from a frame's first kernel on count -- the
lanes' prefills run through the same weight-stationary GEMM kernels as the lock-step frame and used to be summed into it (the round-4
review: 36.9 GB per frame reported where ~11 GB was true)."""
import importlib.util
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench_for_test", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_only_dispatches_from_the_first_frame_kernel_on_are_counted():
    b = _bench()
    per = [("setup_kernel", 1, 5.0), ("skinny_gemm_kernel<1024>", 2, 100.0), ("skinny_gemm_kernel<1024>", 3, 100.0),     # prefill
           ("frame_begin_batch_kernel", 4, 1.0), ("skinny_gemm_kernel<1024>", 5, 7.0), ("attn_decode_lane_kernel", 6, 3.0),
           ("frame_begin_batch_kernel", 7, 1.0), ("skinny_gemm_kernel<1024>", 8, 7.0)]
    rows = dict((n, (c, s)) for n, c, s in b.frame_rows(per, "frame_begin_batch_kernel"))
    assert rows["skinny_gemm_kernel<1024>"] == (2, 14.0)          # the two prefill dispatches (200 KB) are not frame traffic
    assert rows["frame_begin_batch_kernel"] == (2, 2.0) and "setup_kernel" not in rows
    assert b.frame_rows(per, "frame_begin_kernel") == []          # (the single-stream name does not match the batch kernel)
