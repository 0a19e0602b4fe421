"""sdr_amd -- Python host-side mirror of libsdr_amd.so (include/sdr_amd.h).

The product is the C-ABI shared library built from sdr_amd/csrc (hand-written
gfx950 HIP kernels) and the C++ blocks in blocks/.  This package is the thin
ctypes layer tests and bench.py use to reach it; names follow the reference's
own vocabulary for the path:

    FirState(taps, complex_samples)   ~ kfr::fir_state<float, U>   (libs/kfr/dsp/fir.hpp:57-68)
    fir(state, x)                     ~ kfr::fir(state, x)         (libs/kfr/dsp/fir.hpp:167-171)
    hilbert(n)                        ~ sdr::hilbert<float>(n)     (include/sdr/hilbert.hpp:76-81)
    fir_lowpass/... (n, f)            ~ kfr::fir_lowpass/...       (libs/kfr/dsp/fir_design.hpp)

There is no CPU compute path: without the built library or without a GPU the
compute calls raise.
"""
from ._capi import (SdrError, FirState, fir, hilbert, fir_lowpass, fir_highpass, fir_bandpass,
                    fir_bandstop, fir_design, lib, lib_path, device_count, version,
                    host_alloc, host_free, SDR_REAL, SDR_COMPLEX, Chain, Gen, optimal_block_size, WAVEFORMS,
                    fir_apply_device_noalloc)

__all__ = ["SdrError", "FirState", "fir", "hilbert", "fir_lowpass", "fir_highpass", "fir_bandpass",
           "fir_bandstop", "fir_design", "lib", "lib_path", "device_count", "version",
           "host_alloc", "host_free", "SDR_REAL", "SDR_COMPLEX", "Chain", "Gen", "optimal_block_size", "WAVEFORMS",
           "fir_apply_device_noalloc"]
