"""Host-side pieces added late in round 6 (no GPU): the padded-row helper of the feature cache, the kernel-timeline parser of
tools/minibatch_timeline.py, the staging defaults of the loading stage."""
import os
import subprocess
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_padded_rows_views_a_buffer_of_the_asked_pitch():
    from dgll_amd.cache import _padded_rows

    x = torch.arange(5 * 602, dtype=torch.float32).view(5, 602).to(torch.bfloat16)       # 1204-byte rows
    for align, ld in ((1, 602), (16, 608), (128, 640)):
        y = _padded_rows(x, align)
        assert y.shape == x.shape and y.stride() == (ld, 1) and torch.equal(y, x)
        if ld != 602:
            whole = y.as_strided((5, ld), (ld, 1))
            assert bool((whole[:, 602:] == 0).all())                                   # the padding is zero
    z = torch.ones(3, 64, dtype=torch.bfloat16)                                          # already whole lines: a plain copy
    assert _padded_rows(z, 128).stride() == (64, 1)
    assert _padded_rows(x, 3).stride() == (602, 1)                                       # an alignment the element size does not divide


def test_minibatch_timeline_parses_a_kernel_trace(tmp_path):
    """tools/minibatch_timeline.py on a synthetic rocprofv3 kernel trace: 100 batches of 1 ms, a 0.4 ms reduction per batch on queue 3
    beside a 0.7 ms step on queue 2."""
    d = tmp_path / "run"
    d.mkdir()
    rows = ['"Kind","Agent_Id","Queue_Id","Stream_Id","Thread_Id","Dispatch_Id","Kernel_Id","Kernel_Name","Correlation_Id","Start_Timestamp","End_Timestamp"']
    n = 0
    for b in range(100):
        t0 = b * 1_000_000
        for q, name, a, e in ((3, "void dgll::aggregate_rows_kernel<unsigned short, 4>(dgll::AggregateArgs)", 0, 400_000),
                              (3, "upload_kernel(char const*, char*, unsigned long)", 400_000, 500_000),
                              (2, "void dgll::gemm_bf16_nt_kernel<8>(dgll::MfmaGemmArgs)", 100_000, 800_000)):
            n += 1
            rows.append('"KERNEL_DISPATCH","Agent 2",%d,0,1,%d,7,"%s",%d,%d,%d' % (q, n, name, n, t0 + a, t0 + e))
    (d / "1_kernel_trace.csv").write_text("\n".join(rows) + "\n")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "minibatch_timeline.py"), str(tmp_path)], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    text = out.stdout
    assert "1.000 ms per batch" in text and "queue 2      busy 0.700 ms per batch" in text and "queue 3      busy 0.500 ms per batch" in text
    assert "zero-copy kernels (outermost-hop reduction, uploads) busy 0.500 ms per batch" in text
    assert "all queues together busy 0.800 ms per batch" in text


def test_loading_stage_defaults():
    from dgll_amd import pipeline as pl

    # the uncached rows are staged by default, by the library's grid (0), two alternating loading streams are the constructor's default
    assert pl.STAGE_MISSES is True and pl.STAGE_BLOCKS == 0 and pl.STAGE_CAP == 0 and pl.UPLOAD_BLOCKS_ALONE == 128
