"""The allocation-free Griffin-Lim entry point (ttsx_griffin_lim_noalloc: tables and padded signals in a static
device array of the library) against the handle-based entry point, through the C ABI: the same kernels on the
same tables, so the waveforms must be bit-identical - on two calls in a row, the second smaller than the first
(the array keeps the first call's contents), for both STFT geometries."""
import ctypes

import pytest
import torch

pytestmark = pytest.mark.gpu


def _pair(lib, _lib, sp, handle, B, F, iters, preemph, seed, n_frames=None):
    gen = torch.Generator(device="cuda").manual_seed(seed)
    mag = torch.rand(B, F, 1025, device="cuda", generator=gen) + 0.05
    ph = torch.rand(B, F, 1025, device="cuda", generator=gen)
    nf = None if n_frames is None else torch.tensor(n_frames, dtype=torch.int32, device="cuda")
    ref = torch.full((B, 275 * (F - 1)), 3.0, device="cuda")
    got = torch.full((B, 275 * (F - 1)), -3.0, device="cuda")
    st = _lib.current_stream()
    _lib.check(lib.ttsx_vocoder_griffin_lim(handle, _lib.ptr(mag), _lib.ptr(ph), _lib.ptr(nf), B, F, iters, preemph,
                                            _lib.ptr(ref), st), "ttsx_vocoder_griffin_lim")
    _lib.check(lib.ttsx_griffin_lim_noalloc(_lib.ptr(mag), _lib.ptr(ph), _lib.ptr(nf), B, F, ctypes.byref(sp), iters,
                                            preemph, _lib.ptr(got), st), "ttsx_griffin_lim_noalloc")
    torch.cuda.synchronize()
    return ref, got


@pytest.mark.parametrize("win,pad", [(1100, 0), (1102, 1)])
def test_noalloc_griffin_lim_equals_the_handle_entry_point(win, pad):
    from tts_amd import _lib
    lib = _lib.load()
    sp = _lib.StftParams(2048, 275, win, pad)
    h = ctypes.c_void_p()
    _lib.check(lib.ttsx_vocoder_create(ctypes.byref(sp), _lib.current_stream(), ctypes.byref(h)), "create")
    try:
        # first call: 3 ragged utterances, 130 frames, de-emphasis
        ref, got = _pair(lib, _lib, sp, h, 3, 130, 10, 0.98, seed=1, n_frames=[130, 77, 2])
        assert torch.equal(ref, got) and torch.isfinite(got).all() and float(got.abs().max()) > 0
        # second call, smaller: one utterance of 41 frames, no de-emphasis (the last launch writes wav itself)
        ref, got = _pair(lib, _lib, sp, h, 1, 41, 7, 0.0, seed=2)
        assert torch.equal(ref, got) and float(got.abs().max()) > 0
        # and a larger one again, iters = 0 (initial synthesis only)
        ref, got = _pair(lib, _lib, sp, h, 2, 200, 0, 0.98, seed=3)
        assert torch.equal(ref, got)
    finally:
        lib.ttsx_vocoder_destroy(h)


def test_noalloc_rejects_what_does_not_fit_and_allocates_nothing():
    from tts_amd import _lib
    lib = _lib.load()
    sp = _lib.StftParams(2048, 275, 1100, 0)
    one = ctypes.c_void_p(256)
    # 64 x 800 frames = 14.2 M floats per padded signal: beyond the 4 Mi-float static array
    assert lib.ttsx_griffin_lim_noalloc(one, one, None, 64, 800, ctypes.byref(sp), 60, 0.98, one, None) == _lib.ERR_ARG
    assert "static scratch" in lib.ttsx_last_error_string().decode()
    # the largest documented case fits: 16 utterances x 900 frames
    B, F = 16, 900
    mag = torch.rand(B, F, 1025, device="cuda") + 0.05
    ph = torch.rand(B, F, 1025, device="cuda")
    wav = torch.empty(B, 275 * (F - 1), device="cuda")
    # one small call first: the runtime loads a code object, and with it the static array, on its first use,
    # so a process whose first call into this module is the one measured would see that load, not an allocation
    small = torch.empty(1, 275 * 7, device="cuda")
    _lib.check(lib.ttsx_griffin_lim_noalloc(_lib.ptr(mag), _lib.ptr(ph), None, 1, 8, ctypes.byref(sp), 1, 0.98,
                                            _lib.ptr(small), _lib.current_stream()), "ttsx_griffin_lim_noalloc")
    torch.cuda.synchronize()
    before = torch.cuda.mem_get_info()[0]
    for _ in range(2):
        _lib.check(lib.ttsx_griffin_lim_noalloc(_lib.ptr(mag), _lib.ptr(ph), None, B, F, ctypes.byref(sp), 3, 0.98,
                                                _lib.ptr(wav), _lib.current_stream()), "ttsx_griffin_lim_noalloc")
    torch.cuda.synchronize()
    assert torch.cuda.mem_get_info()[0] == before          # free device memory unchanged by the calls
    assert torch.isfinite(wav).all() and float(wav.abs().max()) > 0
