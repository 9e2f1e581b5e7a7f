"""Argument contract of the Python surface (mirrors soxr.resample / ResampleStream: exception
classes and their order, /root/reference/src/soxr/__init__.py:85-131, :182-231) — everything here
is decided on the host before any device call, so it runs without a GPU."""
import numpy as np
import pytest

from conftest import has_gpu


@pytest.mark.parametrize("in_rate,out_rate", [(100, 0), (50, -1), (0, 100.5), (-1.5, 100)])
def test_bad_rates_raise_value_error_first(soxr, in_rate, out_rate):
    with pytest.raises(ValueError):
        soxr.resample(np.zeros(100), in_rate, out_rate)          # float64 input: rates are checked first
    with pytest.raises(ValueError):
        soxr.resample(np.zeros(100, np.complex64), in_rate, out_rate)
    with pytest.raises(ValueError):
        soxr.ResampleStream(in_rate, out_rate, 1)


@pytest.mark.parametrize("dtype", [np.complex64, np.complex128, np.int8, np.int64, np.uint16, np.float16])
def test_bad_dtype_raises_type_error(soxr, dtype):
    with pytest.raises(TypeError):
        soxr.resample(np.zeros(100, dtype), 100, 200)
    with pytest.raises(TypeError):
        soxr.ResampleStream(100, 200, 1, dtype=dtype)
    with pytest.raises(TypeError):
        soxr._resample_oneshot(np.zeros(100, dtype), 100, 200)


@pytest.mark.parametrize("quality", ["best", "", None, 3, 7, 1.5, True])
def test_bad_quality_raises_value_error(soxr, quality):
    with pytest.raises(ValueError):
        soxr.resample(np.zeros(100, np.float32), 100, 200, quality=quality)
    with pytest.raises(ValueError):
        soxr.ResampleStream(100, 200, 1, quality=quality)


def test_quality_names(soxr):
    for q, want in [("VHQ", soxr.VHQ), ("hq", soxr.HQ), ("SOXR_MQ", soxr.MQ), ("lq", soxr.LQ),
                    ("soxr_qq", soxr.QQ), (soxr.VHQ, soxr.VHQ), (np.int64(4), soxr.HQ)]:
        assert soxr._quality_to_enum(q) == want
    assert (soxr.QQ, soxr.LQ, soxr.MQ, soxr.HQ, soxr.VHQ) == (0, 1, 2, 4, 6)


def test_bad_shapes_and_channels(soxr):
    with pytest.raises(ValueError):
        soxr.resample(np.zeros((4, 4, 4), np.float32), 100, 200)
    with pytest.raises(ValueError):
        soxr.resample(np.zeros((4, 0), np.float32), 100, 200)
    with pytest.raises(ValueError):
        soxr.resample(np.zeros((1, 65537), np.float32), 100, 200)
    with pytest.raises(ValueError):
        soxr.ResampleStream(100, 200, 0)
    with pytest.raises(ValueError):
        soxr.ResampleStream(100, 200, 65537)


def test_layout_dispatch_rule(soxr):
    """strides[0] == itemsize -> split-channel path (SURVEY.md §B.2 probe results)."""
    x = np.zeros((100, 2), np.float32)
    assert not soxr._layout_split(x)
    assert soxr._layout_split(np.asfortranarray(x))
    assert soxr._layout_split(np.zeros(100, np.float32))
    assert soxr._layout_split(np.zeros((100, 1), np.float32))
    assert not soxr._layout_split(np.zeros((100, 49), np.float32)[:, :1])
    assert not soxr._layout_split(np.zeros(200, np.float32)[::2])
    assert soxr._layout_split(np.asfortranarray(x)[:0])


def test_module_surface(soxr):
    for name in ("resample", "ResampleStream", "_resample_oneshot", "QQ", "LQ", "MQ", "HQ", "VHQ",
                 "__version__", "__libsoxr_version__"):
        assert hasattr(soxr, name)
    import soxr as alias          # python-soxr_amd/soxr: `import soxr` drop-in
    assert alias.resample is soxr.resample and alias.ResampleStream is soxr.ResampleStream


@pytest.mark.skipif(has_gpu(), reason="checks the no-GPU behaviour")
def test_fails_loudly_without_a_gpu(soxr):
    """No CPU fallback: compute entry points raise instead of silently running elsewhere."""
    with pytest.raises(RuntimeError, match="no HIP device"):
        soxr.resample(np.zeros(100, np.float32), 48000, 44100)
    with pytest.raises(RuntimeError, match="no HIP device"):
        soxr.ResampleStream(48000, 44100, 1)
    with pytest.raises(RuntimeError, match="no HIP device"):
        soxr.resample([0.0] * 10, 48000, 44100)        # list input is coerced to float32 first
