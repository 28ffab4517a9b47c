"""capdec_colsum_static (column sums whose stripe partials live in the library's own fixed device array)
against capdec_colsum (caller workspace), through the C ABI: bit-identical results, two calls in a row with
the second on a smaller input, and a clear refusal past the built-in buffer's size."""
import pytest
import torch

from image_caption_amd import _lib, ops

pytestmark = pytest.mark.gpu


STATIC_BYTES = 4 << 20          # CAPDEC_COLSUM_STATIC_BYTES of include/capdec.h


def _bind():
    return _lib.load()


def _static(lib, x, out):
    M, N = x.shape
    return lib.capdec_colsum_static(x.data_ptr(), _lib.dtype_code(x), M, N, x.stride(0), out.data_ptr(), _lib.stream_ptr())


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_colsum_static_matches_workspace_variant(dtype):
    lib = _bind()
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(11)
    # first call: the largest shape the training step sums (all rows x packed QKV columns at a small batch);
    # second call: fewer rows AND fewer columns, so stale partials of the first call lie beyond what it writes
    for M, N in ((229 * 16, 2304), (77, 517)):
        x = torch.randn(M, N, device=dev, generator=g).to(dtype)
        want = torch.empty(N, dtype=torch.float32, device=dev)
        ops.colsum(x, want)
        got = torch.full((N,), float("nan"), dtype=torch.float32, device=dev)
        rc = _static(lib, x, got)
        assert rc == 0, lib.capdec_last_error()
        torch.cuda.synchronize()
        assert torch.equal(got, want), f"M={M} N={N}: max diff {(got - want).abs().max().item()}"
        ref = x.float().sum(0)
        assert torch.allclose(got, ref, rtol=1e-4, atol=1e-3 * (M ** 0.5))


def test_colsum_static_strided_view_and_empty():
    lib = _bind()
    dev = torch.device("cuda:0")
    x = torch.randn(300, 96, device=dev)
    view = x[:, 32:64]                                  # leading dimension 96, 32 columns
    want = torch.empty(32, dtype=torch.float32, device=dev)
    ops.colsum(view, want)
    got = torch.empty(32, dtype=torch.float32, device=dev)
    assert _static(lib, view, got) == 0
    torch.cuda.synchronize()
    assert torch.equal(got, want)
    got.fill_(7.0)
    # no rows (an empty tensor has no address of its own: any valid pointer stands in): the sums are zero
    assert lib.capdec_colsum_static(x.data_ptr(), _lib.F32, 0, 32, 96, got.data_ptr(), _lib.stream_ptr()) == 0
    torch.cuda.synchronize()
    assert torch.count_nonzero(got).item() == 0


def test_colsum_static_rejects_oversize():
    lib = _bind()
    dev = torch.device("cuda:0")
    cap = STATIC_BYTES
    M, N = 8192, 8192
    assert lib.capdec_colsum_workspace(M, N) > cap
    x = torch.zeros(M, N, dtype=torch.bfloat16, device=dev)
    out = torch.full((N,), 3.0, dtype=torch.float32, device=dev)
    rc = _static(lib, x, out)
    assert rc == -3
    assert b"built-in buffer" in lib.capdec_last_error()
    torch.cuda.synchronize()
    assert bool((out == 3.0).all())                    # nothing was launched


def test_colsum_static_under_graph_capture():
    """What the entry point is for: after one ordinary call on the device (which resolves the array's address), the
    call is recorded into a HIP graph and the graph replayed on fresh inputs, twice."""
    lib = _bind()
    dev = torch.device("cuda:0")
    M, N = 640, 768
    x = torch.randn(M, N, device=dev)
    got = torch.empty(N, dtype=torch.float32, device=dev)
    assert _static(lib, x, got) == 0                    # outside capture: first use on this device
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        with torch.cuda.graph(graph, stream=side):
            rc = _static(lib, x, got)
    assert rc == 0, lib.capdec_last_error()
    want = torch.empty(N, dtype=torch.float32, device=dev)
    for seed in (1, 2):
        x.copy_(torch.randn(M, N, device=dev, generator=torch.Generator(device=dev).manual_seed(seed)))
        got.fill_(float("nan"))
        graph.replay()
        torch.cuda.synchronize()
        ops.colsum(x, want)
        torch.cuda.synchronize()
        assert torch.equal(got, want)
