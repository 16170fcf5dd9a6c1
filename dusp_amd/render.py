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


def write_back(prog, circuit, chunk_size, n_samples, instance=0):
    """Leave the unit objects as the reference leaves them after ticking ceil(n_samples/chunk) chunks: state
    fields from the device (dusp_state_download), circuit.clock advanced (twin of renderChannelData.js writeBack)."""
    D = descriptor
    for u, unit in enumerate(circuit.units):
        op = D.UNITS[type(unit).__name__][0]
        if op in (D.OP_OSC,):
            unit.phase = float(prog.state(u, instance)[0])
        elif op == D.OP_RAMP:
            st = prog.state(u, instance)
            unit.t, unit.playing = float(st[0]), bool(st[1])
        elif op in (D.OP_CB_READER, D.OP_CB_WRITER, D.OP_TIMER):
            unit.t = float(prog.state(u, instance)[0])
        elif op in (D.OP_FIXED_DELAY, D.OP_COMB_FILTER, D.OP_ALL_PASS, D.OP_READBACK_DELAY):
            unit.tBuffer = float(prog.state(u, instance)[0])
        elif op == D.OP_MULTI_OSC:
            st = prog.state(u, instance)
            unit.phase = [float(v) for v in st[1:1 + int(st[0])]]
        elif op == D.OP_FILTER:
            st = prog.state(u, instance)
            if st[0]:
                unit.lastF = float(st[1])
            unit.a0, unit.a1, unit.a2, unit.b1, unit.b2 = (float(v) for v in st[2:7])
            nch = int(st[7])
            hist = st[8:8 + 4 * nch].reshape(nch, 4)
            unit.x1, unit.x2, unit.y1, unit.y2 = ([float(v) for v in hist[:, k]] for k in range(4))
        elif op == D.OP_SHAPE:
            st = prog.state(u, instance)
            unit.t, unit.playing, unit.finished = float(st[0]), bool(st[1]), bool(st[2])
        elif op == D.OP_AHD:
            st = prog.state(u, instance)
            unit.state, unit.playing, unit.t = int(st[0]), bool(st[1]), float(st[2])
        elif op == D.OP_SAMPLE_RATE_REDUX:
            st = prog.state(u, instance)
            unit.timeSinceLastUpdate = float(st[0])
            unit.val = [float(v) for v in st[2:2 + int(st[1])]]
    circuit.clock += ((n_samples + chunk_size - 1) // chunk_size) * chunk_size


def _host_inputs(sources, clock, length):
    """This segment's samples of the circuit's HostSource units -> float32 [n_sources, 1, length] (zeros past a source's end)."""
    if not sources:
        return None
    block = np.zeros((len(sources), 1, length), dtype=np.float32)
    for k, src in enumerate(sources):
        have = src.samples[clock:clock + length]
        block[k, 0, :have.size] = have
    return block


def renderChannelData(outlet, duration=1, TypedArray=np.float32, engine=runtime.ENGINE_AUTO, device=-1):
    """Drop-in for reference src/renderChannelData.js:5-49.  Scheduled events (unit.schedule / scheduleTrigger) are
    honoured the way the reference's Circuit.tick does (Circuit.js:23,57-65): every event due before the end of a
    chunk runs on the host objects before that chunk is rendered; the render is segmented at those boundaries and
    ONE device program is continued from segment to segment (dusp_program_continue), so delay lines, CircleBuffers
    and feedback chunks stay resident on the device.  Afterwards the circuit is consumed like the reference's:
    circuit.clock has advanced and the unit objects hold their post-render state."""
    first = descriptor.extract(outlet, allow_events=True)
    circuit, chunk = first.circuit, first.chunk_size
    n = _n_samples(duration, first.sample_rate)
    result = ChannelData()
    result.sampleRate = first.sample_rate
    if n == 0:
        return result
    has_events = bool(circuit.events)
    ctx = context(first.sample_rate, device)
    end = ((n + chunk - 1) // chunk) * chunk
    prog, clock, segments = None, 0, []
    try:
        while clock < end:
            nxt = end
            if has_events:
                circuit.runEvents(clock + chunk)
                if circuit.events:
                    due = int(circuit.events[0].t // chunk) * chunk
                    nxt = min(end, max(clock + chunk, due))
            ex = first if (clock == 0 and not has_events) else descriptor.extract(outlet, allow_events=True, allow_clock=True)
            if prog is None:
                prog = ctx.build(ex.words, engine | runtime.ENGINE_RESUMABLE if has_events else engine)
            else:
                prog.continue_with(ex.words)
            length = min(nxt, n) - clock  # the last segment may end inside a chunk
            segments.append(prog.render(length, 1, inputs=_host_inputs(ex.sources, clock, length))[0])
            write_back(prog, circuit, chunk, length)  # advances circuit.clock to `nxt`
            clock = nxt
    finally:
        if prog is not None:
            prog.close()
    n_channels = max(seg.shape[0] for seg in segments)
    pcm = np.zeros((n_channels, n), dtype=np.float32)  # late channels start as zeros (renderChannelData.js:38-39)
    at = 0
    for seg in segments:
        pcm[:seg.shape[0], at:at + seg.shape[1]] = seg
        at += seg.shape[1]
    for c in range(n_channels):
        result.append(pcm[c].astype(TypedArray, copy=False))
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
