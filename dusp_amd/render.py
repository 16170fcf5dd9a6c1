"""`renderChannelData(outlet, duration)` — the drop-in surface of the render path
(reference src/renderChannelData.js:5-49), executed on the MI355X.

    channelData = renderChannelData(unit_or_outlet, duration)
    channelData[c]          -> numpy float32 array of duration*sampleRate samples
    channelData.sampleRate  -> like the reference's expando (renderChannelData.js:47)

plus the batched form the reference cannot express: `render_many(outlets, duration)`
renders N structurally identical circuits (voices, a parameter sweep) as ONE
program with a per-instance parameter table.
"""
import numpy as np

from . import descriptor, runtime

_contexts = {}


def context(sample_rate, device=-1):
    key = (device, sample_rate)
    if key not in _contexts:
        _contexts[key] = runtime.Context(device, sample_rate)
    return _contexts[key]


class ChannelData(list):
    """Array of per-channel sample arrays with a `.sampleRate`, as the reference returns."""
    sampleRate = None


def _n_samples(duration, sample_rate):
    n = int(duration * sample_rate)  # `new TypedArray(lengthInSamples)` truncates (renderChannelData.js:24,39)
    if n < 0:
        raise descriptor.DuspError("negative duration")
    return n


def renderChannelData(outlet, duration=1, TypedArray=np.float32, engine=runtime.ENGINE_AUTO, device=-1):
    ex = descriptor.extract(outlet)
    n = _n_samples(duration, ex.sample_rate)
    result = ChannelData()
    result.sampleRate = ex.sample_rate
    if n == 0:
        return result
    prog = context(ex.sample_rate, device).build(ex.words, engine)
    try:
        pcm = prog.render(n, 1)
    finally:
        prog.close()
    ex.circuit.clock = ((n + ex.chunk_size - 1) // ex.chunk_size) * ex.chunk_size  # the circuit has been consumed
    for c in range(pcm.shape[1]):
        result.append(pcm[0, c].astype(TypedArray, copy=False))
    return result


def render_many(outlets, duration=1, engine=runtime.ENGINE_AUTO, device=-1):
    """Render N isomorphic circuits at once -> float32 [N, n_channels, n_samples]."""
    uni = descriptor.unify([descriptor.extract(o) for o in outlets])
    n = _n_samples(duration, uni.sample_rate)
    prog = context(uni.sample_rate, device).build(uni.words, engine)
    try:
        return prog.render(n, uni.n_instances, uni.params)
    finally:
        prog.close()
