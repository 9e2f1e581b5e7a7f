"""`import soxr` drop-in: put python-soxr_amd/ on sys.path and existing python-soxr user code
(e.g. `soxr.resample(audio, sr_in, sr_out, quality="HQ")`) runs on the MI355X path unchanged."""
from soxr_amd import *  # noqa: F401,F403
from soxr_amd import (QQ, LQ, MQ, HQ, VHQ, ResampleStream, resample, _resample_oneshot,  # noqa: F401
                      _resample_divided, __version__, __libsoxr_version__)
