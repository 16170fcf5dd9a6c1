"""dusp_amd — MI355X-native offline render path for Dusp graphs.

Host side (this package): graph construction with the reference's constructor
signatures (graph.py), extraction to a flat descriptor (descriptor.py), and the
ctypes binding of the HIP library (runtime.py).  Device side: dusp_amd/csrc/.
"""
from . import config, descriptor, quick, runtime  # noqa: F401
from .descriptor import DuspError  # noqa: F401
from .graph import (Abs, AllPass, Circuit, CircleBuffer, CircleBufferReader, CircleBufferWriter, Clip, CombFilter, DecibelToScaler, Delay,  # noqa: F401
                    Divide, Filter, FixedDelay, FixedMultiply, Gain, MonoDelay, MultiChannelOsc, ReadBackDelay, HardClipAbove, HardClipBelow, Multiply, Osc, PolarityInvert, Pow,
                    Ramp, Repeater, SecondsToSamples, SemitoneToRatio, Subtract, Sum, Unit,
                    AHD, ConcatChannels, CrossFader, HostSource, MidiToFrequency, Pan, PickChannel, Rescale, SampleRateRedux, Shape, Timer, VectorMagnitude)
from .render import ChannelData, render_many, renderChannelData  # noqa: F401
from .runtime import Context, DuspHipError, Program  # noqa: F401


def configure(sample_rate=None):
    """Set the process-wide sample rate (the reference does this with `--sampleRate=` on argv,
    src/config.js:1,17).  Call before building graphs."""
    if sample_rate is not None:
        config.sampleRate = int(sample_rate)
