"""Graph extractor: live unit graph -> flat descriptor words (float64).

Python twin of dusp_amd/js/lib/extract.js (same words for the same graph;
tests/test_descriptor.py compares the two byte for byte).  Layout: DESIGN.md §3.

The unit ORDER is never re-derived here: it is whatever `circuit.units` holds,
i.e. the result of the reference's own ordering algorithm
(src/Unit.js:171-209 + src/Circuit.js:125-131, restated in dusp_amd/graph.py).
"""
import numpy as np

MAGIC = 1146442576  # 'DUSP'
VERSION = 1
HEADER_WORDS = 12

OP_OSC, OP_RAMP, OP_MULTIPLY, OP_SUM, OP_FILTER, OP_DELAY, OP_CB_READER, OP_CB_WRITER, OP_REPEATER = range(1, 10)
(OP_SUBTRACT, OP_DIVIDE, OP_POLARITY_INVERT, OP_ABS, OP_CLIP, OP_HARD_CLIP_ABOVE, OP_HARD_CLIP_BELOW, OP_SECONDS_TO_SAMPLES,
 OP_FIXED_MULTIPLY, OP_GAIN, OP_DECIBEL_TO_SCALER, OP_SEMITONE_TO_RATIO, OP_POW) = range(10, 23)  # elementwise maps
OP_FIXED_DELAY, OP_COMB_FILTER, OP_ALL_PASS, OP_MONO_DELAY, OP_READBACK_DELAY, OP_MULTI_OSC = range(23, 29)  # §8f-2
(OP_PAN, OP_MIDI_TO_FREQUENCY, OP_RESCALE, OP_CROSS_FADER, OP_VECTOR_MAGNITUDE, OP_TIMER, OP_SAMPLE_RATE_REDUX,
 OP_CONCAT_CHANNELS, OP_PICK_CHANNEL) = range(29, 38)  # rest of §8f-1
OP_SHAPE, OP_AHD = 38, 39  # envelopes (§8f-3)
OP_HOST_ONLY = 40  # no signal: the unit acts through host callbacks between segments (Retriggerer)
OP_INPUT = 41  # a signal the host computes (graph.HostSource; the JS host's Noise): attribute = stream index
OP_RETRIGGER = 42  # a Retriggerer of a Shape / AHD / Ramp, ticked on the device (built by the JS host's extractor)
IN_CONST, IN_CONNECT, IN_PARAM = 0, 1, 2
FILTER_KINDS = {"LP": 0, "HP": 1}

UNITS = {
    "Osc": (OP_OSC, ["f"]),
    "Ramp": (OP_RAMP, []),
    "Multiply": (OP_MULTIPLY, ["a", "b"]),
    "Sum": (OP_SUM, ["a", "b"]),
    "Filter": (OP_FILTER, ["in", "f"]),
    "Delay": (OP_DELAY, ["in", "delay"]),
    "CircleBufferReader": (OP_CB_READER, ["offset"]),
    "CircleBufferWriter": (OP_CB_WRITER, ["offset", "in"]),
    "Repeater": (OP_REPEATER, ["in"]),
    "Subtract": (OP_SUBTRACT, ["a", "b"]),
    "Divide": (OP_DIVIDE, ["a", "b"]),
    "PolarityInvert": (OP_POLARITY_INVERT, ["in"]),
    "Abs": (OP_ABS, ["in"]),
    "Clip": (OP_CLIP, ["in", "threshold"]),
    "HardClipAbove": (OP_HARD_CLIP_ABOVE, ["in", "threshold"]),
    "HardClipBelow": (OP_HARD_CLIP_BELOW, ["in", "threshold"]),
    "SecondsToSamples": (OP_SECONDS_TO_SAMPLES, ["in"]),
    "FixedMultiply": (OP_FIXED_MULTIPLY, ["in"]),
    "Gain": (OP_GAIN, ["in", "gain"]),
    "DecibelToScaler": (OP_DECIBEL_TO_SCALER, ["in"]),
    "SemitoneToRatio": (OP_SEMITONE_TO_RATIO, ["in"]),
    "Pow": (OP_POW, ["a", "b"]),
    "FixedDelay": (OP_FIXED_DELAY, ["in"]),
    "CombFilter": (OP_COMB_FILTER, ["in", "feedbackGain"]),
    "AllPass": (OP_ALL_PASS, ["in", "feedbackGain"]),
    "MonoDelay": (OP_MONO_DELAY, ["in", "delay"]),
    "ReadBackDelay": (OP_READBACK_DELAY, ["in", "delay"]),
    "MultiChannelOsc": (OP_MULTI_OSC, ["f"]),
    "Pan": (OP_PAN, ["in", "pan"]),
    "MidiToFrequency": (OP_MIDI_TO_FREQUENCY, ["midi"]),
    "Rescale": (OP_RESCALE, ["in", "inLower", "inUpper", "outLower", "outUpper"]),
    "CrossFader": (OP_CROSS_FADER, ["a", "b", "dial"]),
    "VectorMagnitude": (OP_VECTOR_MAGNITUDE, ["in"]),
    "Timer": (OP_TIMER, []),
    "SampleRateRedux": (OP_SAMPLE_RATE_REDUX, ["in", "ammount"]),
    "ConcatChannels": (OP_CONCAT_CHANNELS, ["a", "b"]),
    "PickChannel": (OP_PICK_CHANNEL, ["in", "c"]),
    "Shape": (OP_SHAPE, ["duration", "min", "max"]),
    "AHD": (OP_AHD, ["attack", "hold", "decay"]),
    "HostSource": (OP_INPUT, []),
}
DATA_OUTLET = {"MidiToFrequency": "frequency"}  # every other unit's data outlet is "out" (MidiToFrequency.js:6)


def data_outlet_name(unit):
    return DATA_OUTLET.get(type(unit).__name__, "out")


class DuspError(Exception):
    """Raised with the reference's own message strings where it has them."""


class Extraction:
    def __init__(self, words, const_sites, labels, sample_rate, chunk_size, circuit, sources=()):
        self.sources = list(sources)  # HostSource units in stream order (their `samples` feed the render's inputs)
        self.words = words
        self.const_sites = const_sites  # [(kind_pos, val_pos, n)]
        self.labels = labels
        self.sample_rate = sample_rate
        self.chunk_size = chunk_size
        self.circuit = circuit


def to_outlet(x):
    # same checks and messages as reference src/renderChannelData.js:12-17
    if x is None:
        raise DuspError("renderAudioBuffer expects an outlet")
    if getattr(x, "isUnit", False) or getattr(x, "isPatch", False):
        x = x.defaultOutlet
    if not getattr(x, "isOutlet", False):
        raise DuspError("renderAudioBuffer expects an outlet")
    return x


def extract(target, allow_events=False, allow_clock=False):
    """Flatten the circuit of `target` as it stands NOW.  allow_events / allow_clock are for the event-segmented
    render loop (render.py), which extracts once per segment."""
    from .wavetables import WAVEFORMS

    outlet = to_outlet(target)
    out_unit = outlet.unit
    circuit = out_unit.circuit if out_unit.circuit is not None else out_unit.getOrBuildCircuit()
    if circuit.events and not allow_events:
        raise DuspError("dusp-hip: this circuit has scheduled events; extract() flattens a single segment "
                        "(renderChannelData segments them)")
    if getattr(circuit, "promises", None):
        raise DuspError("dusp-hip: circuits with pending promises are not supported on the GPU path")
    if circuit.clock and not allow_clock:
        raise DuspError("dusp-hip: circuit has already been ticked (clock=%s); render a fresh circuit" % circuit.clock)

    units = circuit.units
    chunk = outlet.chunkSize
    rings = []

    def ring_id(buf):
        for i, r in enumerate(rings):
            if r is buf:
                return i
        rings.append(buf)
        return len(rings) - 1

    body, sites, labels, sources = [], [], [], []
    for unit in units:
        kind = type(unit).__name__
        if kind not in UNITS:
            raise DuspError("dusp-hip: unit type not supported on the GPU path: %s (%s)" % (kind, unit.label))
        op, inlet_names = UNITS[kind]
        if unit.tickInterval != chunk:
            raise DuspError("dusp-hip: unit %s has tickInterval %s != chunk size %s" % (unit.label, unit.tickInterval, chunk))
        labels.append(unit.label)
        attrs, state = [], []
        if op == OP_OSC:
            attrs, state = [WAVEFORMS[unit.waveform]], [unit.phase]
        elif op == OP_RAMP:
            attrs, state = [unit.duration, unit.y0, unit.y1], [unit.t, 1 if unit.playing else 0]
        elif op == OP_FILTER:
            if unit.kind not in FILTER_KINDS:
                raise DuspError("dusp-hip: filter kind not supported on the GPU path: %s" % unit.kind)
            attrs = [FILTER_KINDS[unit.kind]]
            nch = max(len(unit.x1), len(unit.x2), len(unit.y1), len(unit.y2))
            has = unit.lastF is not None
            state = [1 if has else 0, unit.lastF if has else 0, unit.a0, unit.a1, unit.a2, unit.b1, unit.b2, nch]
            for c in range(nch):
                state += [unit.x1[c] or 0, unit.x2[c] or 0, unit.y1[c] or 0, unit.y2[c] or 0]
        elif op == OP_DELAY:
            attrs = [unit.maxDelay]
        elif op == OP_CB_READER:
            attrs, state = [ring_id(unit.buffer), 1 if unit.postWipe else 0], [unit.t]
        elif op == OP_CB_WRITER:
            attrs, state = [ring_id(unit.buffer), 1 if unit.preWipe else 0], [unit.t]
        elif op == OP_FIXED_MULTIPLY:
            attrs = [float(unit.sf)]  # a plain number on the unit, not an inlet (FixedMultiply.js:8,20)
        elif op in (OP_FIXED_DELAY, OP_COMB_FILTER, OP_ALL_PASS):
            attrs, state = [unit.delayTimeInSamples], [unit.tBuffer]
        elif op == OP_MONO_DELAY:
            attrs = [unit.maxDelay]
        elif op == OP_READBACK_DELAY:
            attrs, state = [unit.bufferLength], [unit.tBuffer]
        elif op == OP_MULTI_OSC:
            attrs, state = [WAVEFORMS[unit.waveform]], [len(unit.phase)] + [p or 0 for p in unit.phase]
        elif op == OP_SHAPE:
            from .wavetables import SHAPES

            def edge(e):  # the string "shape" (= the table's end value) or a plain number (Shape/index.js:35-49)
                if isinstance(e, str):
                    if e != "shape":
                        raise DuspError('dusp-hip: Shape edge must be "shape" or a number (%s)' % unit.label)
                    return [1, 0]
                return [0, float(e)]
            if getattr(unit, "_finish", None) or getattr(unit, "onFinish", None):
                raise DuspError("dusp-hip: finish callbacks cannot run on the GPU path (%s)" % unit.label)
            attrs = [SHAPES[unit.shape]] + edge(unit.leftEdge) + edge(unit.rightEdge)
            state = [unit.t, 1 if unit.playing else 0, 1 if unit.finished else 0]
        elif op == OP_AHD:
            attrs, state = [1 / outlet.sampleRate], [unit.state, 1 if unit.playing else 0, unit.t]
        elif op == OP_PAN:
            attrs = [float(unit.compensationDB)]  # a plain property (Pan.js:12)
        elif op == OP_TIMER:
            attrs, state = [unit.samplePeriod], [unit.t]
        elif op == OP_INPUT:
            attrs = [len(sources)]
            sources.append(unit)
        elif op == OP_SAMPLE_RATE_REDUX:
            state = [unit.timeSinceLastUpdate, len(unit.val)] + list(unit.val)
        body += [op, len(inlet_names), len(attrs), len(state)]
        for name in inlet_names:
            inlet = unit.inlets[name]
            if inlet.connected:
                src_unit = inlet.outlet.unit
                if src_unit not in units:
                    raise DuspError("dusp-hip: inlet %s is fed from outside the circuit" % inlet.label)
                if inlet.outlet.name != data_outlet_name(src_unit):
                    raise DuspError('dusp-hip: only the data outlet ("out") of a unit can feed an inlet on the GPU path (%s)' % inlet.outlet.label)
                body += [IN_CONNECT, 3, units.index(src_unit), 0, 0]
            else:
                vals = list(inlet.values)
                sites.append((len(body), len(body) + 2, len(vals)))
                body += [IN_CONST, len(vals)] + vals
        body += attrs + state

    if outlet.name != data_outlet_name(out_unit):
        raise DuspError('dusp-hip: only "out" outlets can be rendered on the GPU path')
    ring_words = []
    for r in rings:
        ring_words += [r.numberOfChannels, r.lengthInSamples]
    head = [MAGIC, VERSION, outlet.sampleRate, chunk, len(units), len(rings), 0, units.index(out_unit), 0,
            circuit.clock or 0, 0, 0]
    base = HEADER_WORDS + len(ring_words)
    words = np.array(head + ring_words + body, dtype=np.float64)
    sites = [(k + base, v + base, n) for (k, v, n) in sites]
    return Extraction(words, sites, labels, outlet.sampleRate, chunk, circuit, sources)


class Unified:
    def __init__(self, words, params, n_params, n_instances, extraction):
        self.words = words
        self.params = params  # float32 [n_params, n_instances] (slot-major) or None
        self.n_params = n_params
        self.n_instances = n_instances
        self.sample_rate = extraction.sample_rate
        self.chunk_size = extraction.chunk_size
        self.labels = extraction.labels


def _same(a, b):
    return a == b or (a != a and b != b)


def unify(extractions):
    """N structurally identical circuits (voices / a parameter sweep) -> ONE program plus
    a per-instance parameter table: every unconnected inlet whose constant differs
    between instances becomes a PARAM inlet (twin of extract.js `unify`)."""
    n = len(extractions)
    if n == 0:
        raise DuspError("dusp-hip: no instances")
    first = extractions[0]
    words = first.words.copy()
    for i, e in enumerate(extractions):
        if e.words.shape != words.shape:
            raise DuspError("dusp-hip: instance %d differs in structure from instance 0" % i)
    all_words = np.stack([e.words for e in extractions])
    is_const_val = np.zeros(words.size, dtype=bool)
    for (_, vpos, cnt) in first.const_sites:
        is_const_val[vpos:vpos + cnt] = True
    same = (all_words == words) | (np.isnan(all_words) & np.isnan(words))
    struct_bad = ~same[:, ~is_const_val].all(axis=1)
    if struct_bad.any():
        raise DuspError("dusp-hip: instance %d differs from instance 0 outside inlet constants" % int(np.argmax(struct_bad)))
    columns = []
    for (kpos, vpos, cnt) in first.const_sites:
        if same[:, vpos:vpos + cnt].all():
            continue
        words[kpos] = IN_PARAM
        for k in range(cnt):
            words[vpos + k] = len(columns)
            columns.append(all_words[:, vpos + k].astype(np.float32))
    words[6] = len(columns)
    params = np.ascontiguousarray(np.stack(columns)) if columns else None
    return Unified(words, params, len(columns), n, first)


def single(extraction):
    return Unified(extraction.words, None, 0, 1, extraction)


def state_sites(words):
    """[(first_word, n_words)] of every unit's state block inside a descriptor (layout: DESIGN.md §3)."""
    words = np.asarray(words, dtype=np.float64)
    n_units, n_rings = int(words[4]), int(words[5])
    p = HEADER_WORDS + 2 * n_rings
    sites = []
    for _ in range(n_units):
        n_in, n_attr, n_state = int(words[p + 1]), int(words[p + 2]), int(words[p + 3])
        p += 4
        for _ in range(n_in):
            p += 2 + int(words[p + 1])
        p += n_attr
        sites.append((p, n_state))
        p += n_state
    return sites


def continued(words, clock, states):
    """Descriptor of the same circuit `clock` samples later: the start clock and every unit's state words are
    replaced (states[u] = what Program.state(u) returned after the previous segment; a state block may have
    grown, e.g. a Filter's per-channel history).  This is what a host does between two segments of an
    event-segmented render before calling dusp_program_continue."""
    words = np.asarray(words, dtype=np.float64)
    parts, at = [], 0
    for (first, n), st in zip(state_sites(words), states):
        st = np.asarray(st, dtype=np.float64)
        parts += [words[at:first], st]
        at = first + n
    parts.append(words[at:])
    out = np.concatenate(parts)
    out[9] = clock
    # record headers: n_state of every unit
    p = HEADER_WORDS + 2 * int(out[5])
    for st in states:
        out[p + 3] = len(st)
        n_in, n_attr = int(out[p + 1]), int(out[p + 2])
        p += 4
        for _ in range(n_in):
            p += 2 + int(out[p + 1])
        p += n_attr + len(st)
    return out
