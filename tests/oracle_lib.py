"""ctypes binding of oracle/_build/libsdr_oracle.so -- TEST INFRASTRUCTURE ONLY.

The oracle is the CPU restatement of the reference's FIR hot path
(oracle/sdr_oracle.h).  Only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg may import this module; the product (sdr_amd/) never does.
"""
import ctypes as C
import os
import struct
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
LIB_PATH = os.path.join(ORACLE_DIR, "_build", "libsdr_oracle.so")
REF_DIR = os.path.join(ORACLE_DIR, "_ref")

_lib = None


def build():
    """Compile the port (and, when /root/reference exists, oracle/_ref)."""
    subprocess.run(["make", "-C", ORACLE_DIR, "-j4"], check=True,
                   stdout=subprocess.DEVNULL)


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            build()
        L = C.CDLL(LIB_PATH)
        fp = C.POINTER(C.c_float)
        L.sdro_window_blackman.argtypes = [fp, C.c_size_t]
        L.sdro_hilbert_taps.argtypes = [fp, C.c_size_t]
        L.sdro_fir_design.argtypes = [C.c_int, fp, C.c_size_t, C.c_float, C.c_float]
        L.sdro_window.argtypes = [C.c_int, fp, C.c_size_t]
        L.sdro_window.restype = C.c_int
        L.sdro_fir_design_window.argtypes = [C.c_int, C.c_int, fp, C.c_size_t, C.c_float, C.c_float]
        L.sdro_fir_design_window.restype = C.c_int
        L.sdro_window_param.argtypes = [C.c_int, C.c_float, fp, C.c_size_t]
        L.sdro_window_param.restype = C.c_int
        L.sdro_fir_design_window_param.argtypes = [C.c_int, C.c_int, C.c_float, fp, C.c_size_t, C.c_float, C.c_float]
        L.sdro_fir_design_window_param.restype = C.c_int
        L.sdro_fir_create.argtypes = [fp, C.c_size_t, C.c_int, C.c_int]
        L.sdro_fir_create.restype = C.c_void_p
        L.sdro_fir_reset.argtypes = [C.c_void_p]
        L.sdro_fir_destroy.argtypes = [C.c_void_p]
        L.sdro_fir_process.argtypes = [C.c_void_p, fp, C.c_size_t, fp]
        L.sdro_optimal_block_size.argtypes = [C.c_uint64, C.c_uint64]
        L.sdro_optimal_block_size.restype = C.c_size_t
        L.sdro_fastsin.argtypes = [C.c_float]
        L.sdro_fastsin.restype = C.c_float
        L.sdro_gen_real.argtypes = [fp, C.c_size_t, C.c_size_t, C.c_float, C.c_float, fp, C.c_int]
        L.sdro_gen_complex.argtypes = [fp, C.c_size_t, C.c_size_t, C.c_float, C.c_float, fp, C.c_int]
        L.sdro_gen_quad_preroll.argtypes = [C.c_void_p, C.c_size_t, C.c_float, C.c_float, C.c_int]
        L.sdro_gen_complex_quad.argtypes = [fp, C.c_size_t, C.c_size_t, C.c_float, C.c_float, fp, C.c_int,
                                            C.c_void_p, C.c_size_t]
        L.sdro_hilbert_block.argtypes = [C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t,
                                         C.c_uint16, C.c_size_t, C.c_int, C.c_int]
        L.sdro_hilbert_block.restype = C.c_size_t
        L.sdro_stream_filter_forward.argtypes = [
            C.c_int, C.POINTER(C.c_uint64), C.c_size_t, C.c_int,
            C.POINTER(C.c_uint16), C.c_size_t, C.c_int, C.c_uint16, C.c_uint16]
        L.sdro_stream_filter_forward.restype = C.c_int
        _lib = L
    return _lib


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def window_blackman(n):
    w = np.empty(n, np.float32)
    lib().sdro_window_blackman(_fp(w), n)
    return w


def hilbert_taps(n):
    t = np.empty(n, np.float32)
    lib().sdro_hilbert_taps(_fp(t), n)
    return t


FIR_KINDS = {"lowpass": 0, "highpass": 1, "bandpass": 2, "bandstop": 3}


WINDOWS = {"blackman": 0, "rectangular": 1, "hann": 2, "hamming": 3, "blackman_harris": 4, "kaiser": 5}


def fir_design(kind, n, f1, f2=0.0, window="blackman", param=0.5):
    t = np.empty(n, np.float32)
    assert lib().sdro_fir_design_window_param(FIR_KINDS[kind], WINDOWS[window], param, _fp(t), n, f1, f2) == 0
    return t


def window(name, n, param=0.5):
    w = np.empty(n, np.float32)
    assert lib().sdro_window_param(WINDOWS[name], param, _fp(w), n) == 0
    return w


class Fir:
    """Stateful streaming FIR (kfr::fir_state + kfr::fir restated)."""

    def __init__(self, taps, complex_samples, acc64=True):
        taps = np.ascontiguousarray(taps, np.float32)
        self.cplx = bool(complex_samples)
        self.h = lib().sdro_fir_create(_fp(taps), len(taps), int(self.cplx), int(acc64))

    def process(self, x):
        if self.cplx:
            x = np.ascontiguousarray(x, np.complex64)
            y = np.empty_like(x)
            lib().sdro_fir_process(self.h, _fp(x.view(np.float32)), len(x), _fp(y.view(np.float32)))
        else:
            x = np.ascontiguousarray(x, np.float32)
            y = np.empty_like(x)
            lib().sdro_fir_process(self.h, _fp(x), len(x), _fp(y))
        return y

    def reset(self):
        lib().sdro_fir_reset(self.h)

    def __del__(self):
        if getattr(self, "h", None):
            lib().sdro_fir_destroy(self.h)
            self.h = None


def fir(taps, x, acc64=True):
    x = np.asarray(x)
    return Fir(taps, np.iscomplexobj(x), acc64).process(x)


def optimal_block_size(elt, sample_rate=0):
    return lib().sdro_optimal_block_size(elt, sample_rate)


def gen_real(blocks, block_size, amp, cps, phi, waveform):
    out = np.empty(blocks * block_size, np.float32)
    p = C.c_float(phi)
    lib().sdro_gen_real(_fp(out), blocks, block_size, amp, cps, C.byref(p), waveform)
    return out, p.value


def gen_complex(blocks, block_size, amp, cps, phi, waveform):
    out = np.empty(blocks * block_size, np.complex64)
    p = C.c_float(phi)
    lib().sdro_gen_complex(_fp(out.view(np.float32)), blocks, block_size, amp, cps, C.byref(p), waveform)
    return out, p.value


def gen_complex_quad(blocks, block_size, amp, cps, phi, waveform, ntaps=127, preroll=True, acc64=False):
    """Complex square/triangle/sawtooth: waveform + (+-j) * its Hilbert FIR (gen.cpp:231-260)."""
    out = np.empty(blocks * block_size, np.complex64)
    f = Fir(hilbert_taps(ntaps), False, acc64)
    hd = (ntaps - 1) // 2
    if preroll:
        lib().sdro_gen_quad_preroll(f.h, hd, cps, phi, waveform)
    p = C.c_float(phi)
    lib().sdro_gen_complex_quad(_fp(out.view(np.float32)), blocks, block_size, amp, cps, C.byref(p), waveform, f.h, hd)
    return out, p.value


def hilbert_block(stream_bytes, stream_id=0, taps=127, delay=True, emulate_q1=False):
    cap = 2 * len(stream_bytes) + 64
    buf = C.create_string_buffer(cap)
    n = lib().sdro_hilbert_block(stream_bytes, len(stream_bytes), buf, cap,
                                 stream_id, taps, int(delay), int(emulate_q1))
    assert n != C.c_size_t(-1).value
    return buf.raw[:n]


def stream_filter_forward(mode, ids, contents, pkt_id, pkt_content):
    """mode 'pass'|'drop'; ids/contents: None = option unset, else iterable."""
    ia = (C.c_uint64 * max(1, len(ids or [])))(*(ids or []))
    ca = (C.c_uint16 * max(1, len(contents or [])))(*(contents or []))
    return bool(lib().sdro_stream_filter_forward(
        0 if mode == "pass" else 1, ia, len(ids or []), int(ids is not None),
        ca, len(contents or []), int(contents is not None), pkt_id, pkt_content))


# ---- packet helpers (include/sdr/packet.hpp:28-45 layout) -------------------
HDR = struct.Struct("<HHIQ")
SIGNAL, COMPLEX_SIGNAL = 6, 7


def pack_packet(pid, content, payload, duration=0, size=None):
    return HDR.pack(pid, content, len(payload) if size is None else size, duration) + payload


def parse_packets(buf):
    """-> list of (id, content, size, duration, payload bytes actually present)."""
    out, p = [], 0
    while len(buf) - p >= HDR.size:
        pid, content, size, dur = HDR.unpack_from(buf, p)
        p += HDR.size
        payload = buf[p:p + size]
        p += len(payload)
        out.append((pid, content, size, dur, payload))
    return out


def ref_bin(name):
    """Path of a compiled reference binary in oracle/_ref, or None."""
    p = os.path.join(REF_DIR, name)
    return p if os.path.exists(p) else None
