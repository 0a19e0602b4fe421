"""ctypes binding of libsdr_amd.so.  Signatures follow include/sdr_amd.h."""
import ctypes as C
import os

import numpy as np

SDR_REAL, SDR_COMPLEX = 0, 1
_KINDS = {"lowpass": 0, "highpass": 1, "bandpass": 2, "bandstop": 3}

_HERE = os.path.dirname(os.path.abspath(__file__))
_lib = None


class SdrError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("sdr_amd error %d: %s" % (code, msg))
        self.code = code


def lib_path():
    # SDR_AMD_LIB: an alternative build of the library (kernel experiments)
    return os.environ.get("SDR_AMD_LIB") or os.path.join(_HERE, "lib", "libsdr_amd.so")


def lib():
    """Load libsdr_amd.so (built in-tree by `make -C sdr_amd/csrc`).  Fails loudly."""
    global _lib
    if _lib is None:
        path = lib_path()
        if not os.path.exists(path):
            raise ImportError("libsdr_amd.so is not built (%s missing): run `make -C sdr_amd/csrc` "
                              "or __graft_entry__.build(); there is no fallback path" % path)
        L = C.CDLL(path)
        fp, vp, sz = C.POINTER(C.c_float), C.c_void_p, C.c_size_t
        L.sdr_version.restype = C.c_char_p
        L.sdr_last_error.restype = C.c_char_p
        L.sdr_device_count.restype = C.c_int
        L.sdr_hilbert_taps.argtypes = [fp, sz]
        L.sdr_fir_design.argtypes = [C.c_int, fp, sz, C.c_float, C.c_float]
        L.sdr_fir_design_window.argtypes = [C.c_int, C.c_int, C.c_float, fp, sz, C.c_float, C.c_float]
        L.sdr_fir_create.argtypes = [C.POINTER(vp), fp, sz, C.c_int, sz, C.c_int]
        L.sdr_fir_create_numerics.argtypes = [C.POINTER(vp), fp, sz, C.c_int, sz, C.c_int, C.c_int]
        # test hook outside the C ABI (sdr_amd/csrc/internal.h): pins the kernel family
        L.sdr_amd_internal_fir_create.argtypes = [C.POINTER(vp), fp, sz, C.c_int, sz, C.c_int, C.c_int, C.c_int]
        L.sdr_fir_process_device.argtypes = [vp, vp, sz, sz, vp, sz, vp]
        L.sdr_fir_apply_device_noalloc.argtypes = [fp, sz, C.c_int, vp, sz, vp, C.c_int, vp]
        L.sdr_fir_process.argtypes = [vp, vp, sz, vp]
        L.sdr_fir_submit.argtypes = [vp, C.c_int, vp, sz, vp]
        L.sdr_fir_wait.argtypes = [vp, C.c_int]
        L.sdr_fir_reset.argtypes = [vp]
        L.sdr_fir_destroy.argtypes = [vp]
        L.sdr_fir_destroy.restype = None
        L.sdr_fir_ntaps.argtypes = [vp]
        L.sdr_fir_ntaps.restype = sz
        L.sdr_fir_tap_stride.argtypes = [vp]
        L.sdr_fir_kernel_name.argtypes = [vp]
        L.sdr_fir_kernel_name.restype = C.c_char_p
        L.sdr_chain_create.argtypes = [C.POINTER(vp), sz, C.POINTER(fp), C.POINTER(sz), C.POINTER(sz), C.c_int, sz, C.c_int]
        L.sdr_chain_process_device.argtypes = [vp, vp, sz, sz, vp, sz, C.POINTER(sz), vp]
        L.sdr_chain_reset.argtypes = [vp]
        L.sdr_chain_seek.argtypes = [vp, sz]
        L.sdr_chain_min_seek.argtypes = [vp]
        L.sdr_chain_min_seek.restype = sz
        L.sdr_gen_advance.argtypes = [vp, sz]
        L.sdr_gen_set_hilbert.argtypes = [vp, sz, C.c_int]
        L.sdr_chain_destroy.argtypes = [vp]
        L.sdr_chain_destroy.restype = None
        L.sdr_chain_ntaps.argtypes = [vp]
        L.sdr_chain_ntaps.restype = sz
        L.sdr_chain_generate_device.argtypes = [vp, vp, sz, vp, C.POINTER(sz), vp]
        L.sdr_gen_create.argtypes = [C.POINTER(vp), C.c_int, C.c_int, C.c_float, C.c_float, C.c_float, sz, C.c_int]
        L.sdr_gen_set_frequency.argtypes = [vp, C.c_float]
        L.sdr_gen_generate_device.argtypes = [vp, sz, vp, vp]
        L.sdr_gen_generate.argtypes = [vp, sz, vp]
        L.sdr_gen_destroy.argtypes = [vp]
        L.sdr_gen_destroy.restype = None
        L.sdr_optimal_block_size.argtypes = [C.c_uint64, C.c_uint64]
        L.sdr_optimal_block_size.restype = sz
        L.sdr_host_alloc.argtypes = [sz]
        L.sdr_host_alloc.restype = vp
        L.sdr_host_free.argtypes = [vp]
        L.sdr_host_free.restype = None
        _lib = L
    return _lib


def _check(rc):
    if rc != 0:
        raise SdrError(rc, lib().sdr_last_error().decode())


def version():
    return lib().sdr_version().decode()


def device_count():
    return lib().sdr_device_count()


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def hilbert(n):
    """Hilbert-transformer taps, as sdr::hilbert<float>(n) in the reference."""
    t = np.empty(int(n), np.float32)
    _check(lib().sdr_hilbert_taps(_fp(t), int(n)))
    return t


WINDOWS_FIR = {"blackman": 0, "rectangular": 1, "hann": 2, "hamming": 3, "blackman_harris": 4, "kaiser": 5}


def fir_design(kind, n, f1, f2=0.0, window="blackman", beta=0.5):
    """kfr::fir_* style taps; `beta` is the Kaiser window's parameter."""
    t = np.empty(int(n), np.float32)
    _check(lib().sdr_fir_design_window(_KINDS[kind], WINDOWS_FIR[window], beta, _fp(t), int(n), f1, f2))
    return t


def fir_lowpass(n, cutoff, window="blackman", beta=0.5):
    return fir_design("lowpass", n, cutoff, 0.0, window, beta)


def fir_highpass(n, cutoff, window="blackman", beta=0.5):
    return fir_design("highpass", n, cutoff, 0.0, window, beta)


def fir_bandpass(n, f1, f2, window="blackman", beta=0.5):
    return fir_design("bandpass", n, f1, f2, window, beta)


def fir_bandstop(n, f1, f2, window="blackman", beta=0.5):
    return fir_design("bandstop", n, f1, f2, window, beta)


def host_alloc(nbytes):
    p = lib().sdr_host_alloc(nbytes)
    if not p:
        raise SdrError(-4, lib().sdr_last_error().decode())
    return p


def host_free(p):
    lib().sdr_host_free(p)


class FirState:
    """Streaming FIR state on one GPU: real taps, real or complex samples,
    `channels` independent streams.  History is zero after construction and is
    carried from call to call, like kfr::fir_state in the reference."""

    PATHS = {"auto": -1, "direct": 0, "fft": 1, "mfma": 2, "fft4k": 3, "fft16k": 4}
    NUMERICS = {"fast": 0, "per-output": 1}

    def __init__(self, taps, complex_samples=False, channels=1, device=0, path="auto", grid=0, numerics=None):
        """path / grid: tests and experiments only -- pin the kernel family ("direct", "fft" =
        overlap-save) and the persistent grid
        through the library's internal hook; the default goes through sdr_fir_create."""
        taps = np.ascontiguousarray(taps, np.float32)
        self.complex_samples = bool(complex_samples)
        self.channels = int(channels)
        self.device = int(device)
        self._h = C.c_void_p()
        kind = SDR_COMPLEX if self.complex_samples else SDR_REAL
        if numerics is not None:   # the public switch (sdr_fir_create_numerics): "fast" or "per-output"
            _check(lib().sdr_fir_create_numerics(C.byref(self._h), _fp(taps), len(taps), kind, self.channels, self.device,
                                                 self.NUMERICS[numerics] if isinstance(numerics, str) else int(numerics)))
        elif path == "auto" and not grid:
            _check(lib().sdr_fir_create(C.byref(self._h), _fp(taps), len(taps), kind, self.channels, self.device))
        else:
            _check(lib().sdr_amd_internal_fir_create(C.byref(self._h), _fp(taps), len(taps), kind, self.channels,
                                                     self.device, self.PATHS[path], int(grid)))

    # -- introspection
    @property
    def ntaps(self):
        return lib().sdr_fir_ntaps(self._h)

    @property
    def tap_stride(self):
        return lib().sdr_fir_tap_stride(self._h)

    @property
    def kernel_name(self):
        return lib().sdr_fir_kernel_name(self._h).decode()

    @property
    def dtype(self):
        return np.complex64 if self.complex_samples else np.float32

    # -- host buffers
    def process(self, x):
        """x: (n,) or (channels, n) host array -> same shape, filtered."""
        x = np.ascontiguousarray(x, self.dtype)
        n = x.shape[-1]
        if x.size != n * self.channels:
            raise ValueError("expected %d channel(s) of samples" % self.channels)
        y = np.empty_like(x)
        _check(lib().sdr_fir_process(self._h, x.ctypes.data, n, y.ctypes.data))
        return y

    def submit(self, slot, in_ptr, n, out_ptr):
        _check(lib().sdr_fir_submit(self._h, slot, in_ptr, n, out_ptr))

    def wait(self, slot):
        _check(lib().sdr_fir_wait(self._h, slot))

    # -- device buffers
    def process_device(self, d_in, n, d_out, in_stride=None, out_stride=None, stream=None):
        """Raw device pointers (ints).  Asynchronous on `stream` (a hipStream_t value)."""
        _check(lib().sdr_fir_process_device(self._h, d_in, n if in_stride is None else in_stride, n,
                                            d_out, n if out_stride is None else out_stride,
                                            C.c_void_p(stream or 0)))

    def process_tensor(self, x, out=None):
        """torch tensors resident on this state's GPU (torch is only the
        allocator here): float32 (n,) / (channels, n) for real samples, complex64 or
        float32 (..., n, 2) for complex.  Runs on torch's current stream."""
        import torch
        assert x.is_cuda and x.is_contiguous()
        if out is None:
            out = torch.empty_like(x)
        assert out.is_cuda and out.is_contiguous() and out.shape == x.shape and out.dtype == x.dtype
        if self.complex_samples:
            n = x.shape[-1] if x.is_complex() else x.shape[-2]
        else:
            n = x.shape[-1]
        per = x.numel() if x.is_complex() or not self.complex_samples else x.numel() // 2
        if per != n * self.channels:
            raise ValueError("tensor shape does not match %d channel(s)" % self.channels)
        stream = torch.cuda.current_stream(x.device).cuda_stream
        self.process_device(x.data_ptr(), n, out.data_ptr(), stream=stream)
        return out

    def reset(self):
        _check(lib().sdr_fir_reset(self._h))

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            lib().sdr_fir_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def fir_apply_device_noalloc(taps, complex_samples, d_in, n, d_out, device=0, stream=None):
    """sdr_fir_apply_device_noalloc: stateless one-shot FIR on raw device pointers that allocates
    nothing (<= 255 taps; zero history)."""
    taps = np.ascontiguousarray(taps, np.float32)
    _check(lib().sdr_fir_apply_device_noalloc(_fp(taps), len(taps), SDR_COMPLEX if complex_samples else SDR_REAL,
                                              d_in, int(n), d_out, int(device), C.c_void_p(stream or 0)))


def fir(state, x):
    """kfr::fir(state, x): filter `x` through `state`, advancing its history."""
    if hasattr(x, "is_cuda"):
        return state.process_tensor(x)
    return state.process(x)


class Chain:
    """A cascade of FIR stages (each: taps, then drop `skip` leading outputs) run as one
    pass on the GPU -- the fused form of `hilbert | stream-filter fir=...`."""

    def __init__(self, stages, complex_samples=False, channels=1, device=0):
        """stages: list of (taps, skip)."""
        self.complex_samples = bool(complex_samples)
        self.channels = int(channels)
        taps = [np.ascontiguousarray(t, np.float32) for t, _ in stages]
        self._keep = taps
        n = len(stages)
        tp = (C.POINTER(C.c_float) * n)(*[_fp(t) for t in taps])
        nt = (C.c_size_t * n)(*[len(t) for t in taps])
        sk = (C.c_size_t * n)(*[int(s) for _, s in stages])
        self._h = C.c_void_p()
        _check(lib().sdr_chain_create(C.byref(self._h), n, tp, nt, sk,
                                      SDR_COMPLEX if self.complex_samples else SDR_REAL, self.channels, int(device)))

    @property
    def ntaps(self):
        return lib().sdr_chain_ntaps(self._h)

    def process_device(self, d_in, n, d_out, in_stride=None, out_stride=None, stream=None):
        n_out = C.c_size_t(0)
        _check(lib().sdr_chain_process_device(self._h, d_in, n if in_stride is None else in_stride, n, d_out,
                                              n if out_stride is None else out_stride, C.byref(n_out),
                                              C.c_void_p(stream or 0)))
        return n_out.value

    def generate_device(self, gen, blocks, d_out, stream=None):
        """The chain's outputs for the next `blocks` packets of `gen` (a Gen), synthesised
        inside the FIR kernel; returns the number of outputs written at d_out."""
        n_out = C.c_size_t(0)
        _check(lib().sdr_chain_generate_device(self._h, gen._h, blocks, d_out, C.byref(n_out), C.c_void_p(stream or 0)))
        return n_out.value

    def process_tensor(self, x, out=None):
        """x: (n,) / (channels, n) float32 [real] or (..., n, 2) [complex] CUDA tensor.
        Returns (out, n_out): the first n_out samples of each channel row are valid."""
        import torch
        assert x.is_cuda and x.is_contiguous()
        if out is None:
            out = torch.empty_like(x)
        assert out.is_cuda and out.is_contiguous() and out.shape == x.shape and out.dtype == x.dtype
        if self.complex_samples:
            n = x.shape[-1] if x.is_complex() else x.shape[-2]
        else:
            n = x.shape[-1]
        per = x.numel() if x.is_complex() or not self.complex_samples else x.numel() // 2
        if per != n * self.channels:
            raise ValueError("tensor shape does not match %d channel(s)" % self.channels)
        stream = torch.cuda.current_stream(x.device).cuda_stream
        n_out = self.process_device(x.data_ptr(), n, out.data_ptr(), stream=stream)
        return out, n_out

    def reset(self):
        _check(lib().sdr_chain_reset(self._h))

    @property
    def min_seek(self):
        """Smallest non-zero position seek() accepts (end of the start-of-stream region)."""
        return lib().sdr_chain_min_seek(self._h)

    def seek(self, position):
        """Not at the start of the stream: `position` inputs were consumed elsewhere."""
        _check(lib().sdr_chain_seek(self._h, int(position)))

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            lib().sdr_chain_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


WAVEFORMS = {"cosine": 0, "sine": 1, "square": 2, "triangle": 3, "sawtooth": 4}


class Gen:
    """The reference `gen` block's sample synthesis on the GPU (sdr_gen_*)."""

    def __init__(self, waveform, complex_samples, amp, cycles_per_sample, phase_degrees, block_size, device=0):
        self.complex_samples = bool(complex_samples)
        self.block_size = int(block_size)
        self._h = C.c_void_p()
        _check(lib().sdr_gen_create(C.byref(self._h), WAVEFORMS[waveform],
                                    SDR_COMPLEX if self.complex_samples else SDR_REAL, amp, cycles_per_sample,
                                    phase_degrees, self.block_size, int(device)))

    def generate(self, blocks):
        out = np.empty(blocks * self.block_size, np.complex64 if self.complex_samples else np.float32)
        _check(lib().sdr_gen_generate(self._h, blocks, out.ctypes.data))
        return out

    def generate_device(self, blocks, d_out, stream=None):
        _check(lib().sdr_gen_generate_device(self._h, blocks, d_out, C.c_void_p(stream or 0)))

    def set_hilbert(self, hilbert_taps, preroll=True):
        """Quadrature filter of complex square/triangle/sawtooth (gen.cpp:138, 147-164)."""
        _check(lib().sdr_gen_set_hilbert(self._h, int(hilbert_taps), int(bool(preroll))))

    def advance(self, blocks):
        """Skip `blocks` packets (phase recurrence only)."""
        _check(lib().sdr_gen_advance(self._h, int(blocks)))

    def set_frequency(self, cps):
        _check(lib().sdr_gen_set_frequency(self._h, cps))

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            lib().sdr_gen_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def optimal_block_size(element_size, sample_rate=0):
    return lib().sdr_optimal_block_size(element_size, sample_rate)
