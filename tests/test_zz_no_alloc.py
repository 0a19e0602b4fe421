"""The no-allocation entry point (sdr_fir_apply_device_noalloc, include/sdr_amd.h) against the
existing entry points it is a variant of (sdr_fir_create + sdr_fir_process_device on a fresh
handle), through the C ABI: two calls in a row on one GPU, the second with a smaller input --
the fixed scratch array keeps the first call's contents and must be re-initialised."""
import numpy as np
import pytest

import sdr_amd
import oracle_lib as O

pytestmark = pytest.mark.gpu
TOL = 1e-5


def _run_pair(taps, cplx, sizes, seed):
    import torch
    rng = np.random.default_rng(seed)
    w = 2 if cplx else 1
    outs = []
    for n in sizes:
        x = rng.uniform(-1, 1, n * w).astype(np.float32)
        xd = torch.from_numpy(x).cuda()
        yd = torch.full_like(xd, 7.0)
        stream = torch.cuda.current_stream().cuda_stream
        sdr_amd.fir_apply_device_noalloc(taps, cplx, xd.data_ptr(), n, yd.data_ptr(), device=0, stream=stream)
        # the existing entry point on a fresh handle, direct-form kernels (the family the
        # no-allocation variant shares), same stream
        st = sdr_amd.FirState(taps, cplx, path="direct")
        zd = torch.full_like(xd, -7.0)
        st.process_device(xd.data_ptr(), n, zd.data_ptr(), stream=stream)
        torch.cuda.synchronize()
        outs.append((x, yd.cpu().numpy(), zd.cpu().numpy()))
        st.close()
    return outs


@pytest.mark.parametrize("cplx", [False, True])
@pytest.mark.parametrize("kind,ntaps", [("lowpass", 127), ("hilbert", 127), ("lowpass", 31), ("random", 5), ("lowpass", 255)])
def test_no_alloc_matches_process_device_twice_in_a_row(kind, ntaps, cplx):
    if kind == "hilbert":
        taps = sdr_amd.hilbert(ntaps)
    elif kind == "lowpass":
        taps = sdr_amd.fir_lowpass(ntaps, 0.1)
    else:
        taps = np.random.default_rng(ntaps).uniform(-1, 1, ntaps).astype(np.float32)
    # second call smaller than the first (and smaller than the history the first one left)
    for (x, y, z), n in zip(_run_pair(taps, cplx, [70001, 100], ntaps + cplx), [70001, 100]):
        # same kernels, same inputs: bit-identical to the existing entry point
        assert np.array_equal(y, z), "no-alloc output differs from sdr_fir_process_device (n=%d)" % n
        xs = x.view(np.complex64) if cplx else x
        ys = y.view(np.complex64) if cplx else y
        ref = O.fir(taps, xs)
        rms = np.sqrt(np.mean(np.abs(ref.astype(np.complex128)) ** 2))
        assert np.max(np.abs(ys.astype(np.complex128) - ref)) / rms <= TOL


def test_no_alloc_matches_default_path_within_tolerance():
    """Against plain sdr_fir_create (which picks the overlap-save kernels at 127 taps): same
    filter, different arithmetic order -- within the fp32 bar."""
    import torch
    taps = sdr_amd.fir_lowpass(127, 0.1)
    rng = np.random.default_rng(5)
    for n in (50000, 3000):
        x = rng.uniform(-1, 1, 2 * n).astype(np.float32)
        xd = torch.from_numpy(x).cuda()
        yd = torch.empty_like(xd)
        sdr_amd.fir_apply_device_noalloc(taps, True, xd.data_ptr(), n, yd.data_ptr())
        st = sdr_amd.FirState(taps, True)
        zd = st.process_tensor(xd.view(n, 2))
        torch.cuda.synchronize()
        y = yd.cpu().numpy().view(np.complex64)
        z = zd.cpu().numpy().reshape(-1).view(np.complex64)
        rms = np.sqrt(np.mean(np.abs(z.astype(np.complex128)) ** 2))
        assert np.max(np.abs(y.astype(np.complex128) - z)) / rms <= TOL


def test_no_alloc_limits():
    import torch
    xd = torch.zeros(1000, device="cuda")
    yd = torch.zeros(1000, device="cuda")
    with pytest.raises(sdr_amd.SdrError) as e:
        sdr_amd.fir_apply_device_noalloc(np.ones(256, np.float32), False, xd.data_ptr(), 1000, yd.data_ptr())
    assert "SDR_NOALLOC_MAX_TAPS" in str(e.value)
    with pytest.raises(sdr_amd.SdrError):
        sdr_amd.fir_apply_device_noalloc(np.ones(5, np.float32), False, xd.data_ptr(), (1 << 36) + 1, yd.data_ptr())
    with pytest.raises(sdr_amd.SdrError):   # in and out overlap
        sdr_amd.fir_apply_device_noalloc(np.ones(5, np.float32), False, xd.data_ptr(), 1000, xd.data_ptr() + 400)


def test_no_alloc_under_hip_graph_capture():
    """One of the callers the entry point is for: code under HIP graph capture.  After one ordinary call on
    the device (which takes the scratch array's address), a call captured into a graph and replayed on new
    input gives the same bytes as the ordinary call."""
    import torch
    taps = sdr_amd.fir_lowpass(63, 0.2)
    rng = np.random.default_rng(9)
    n = 40000
    xd = torch.from_numpy(rng.uniform(-1, 1, n).astype(np.float32)).cuda()
    yd = torch.empty_like(xd)
    sdr_amd.fir_apply_device_noalloc(taps, False, xd.data_ptr(), n, yd.data_ptr(),
                                     stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    graph = torch.cuda.CUDAGraph()
    gd = torch.empty_like(xd)
    with torch.cuda.graph(graph, stream=side):
        sdr_amd.fir_apply_device_noalloc(taps, False, xd.data_ptr(), n, gd.data_ptr(),
                                         stream=torch.cuda.current_stream().cuda_stream)
    for _ in range(2):
        x = rng.uniform(-1, 1, n).astype(np.float32)
        xd.copy_(torch.from_numpy(x))
        graph.replay()
        sdr_amd.fir_apply_device_noalloc(taps, False, xd.data_ptr(), n, yd.data_ptr(),
                                         stream=torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        assert np.array_equal(gd.cpu().numpy(), yd.cpu().numpy())
        ref = O.fir(taps, x)
        assert np.max(np.abs(gd.cpu().numpy() - ref)) / np.sqrt(np.mean(ref.astype(np.float64) ** 2)) <= TOL
