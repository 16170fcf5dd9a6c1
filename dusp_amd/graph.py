"""Host-side mirror of the reference's graph-construction API for the render path.

Only what `renderChannelData` needs is here: the port/buffer model
(reference src/Piglet.js, Inlet.js, Outlet.js), the Unit base with the
process-index ordering (src/Unit.js:171-209), the Circuit that owns the sorted
unit list (src/Circuit.js:67-148) and the constructors of the units the GPU path
executes (src/components/{Osc/Osc,Ramp,Multiply,Sum,Filter,Delay,
CircleBufferReader,CircleBufferWriter,Repeater}.js, src/CircleBuffer.js).

These classes never tick: there is no CPU implementation of any `_tick` in this
package.  They exist so that graphs can be BUILT with the reference's
constructor signatures / defaults and handed to the extractor
(dusp_amd/descriptor.py), which flattens them for the HIP library.
"""
import contextlib
import functools
import math
import sys

import numpy as np

from . import config


def _f32(v):
    return float(np.float32(v))


@contextlib.contextmanager
def _deep_recursion():
    """The reference's flood fill and process-index walk recurse once per unit along a chain
    (Circuit.js:96-101, Unit.js:185-205); a 1024-voice Sum.many chain is ~2000 frames deep."""
    old = sys.getrecursionlimit()
    sys.setrecursionlimit(max(old, 200000))
    try:
        yield
    finally:
        sys.setrecursionlimit(old)


class _Port:
    """Common part of Inlet and Outlet (reference src/Piglet.js:5-23)."""

    def __init__(self, unit, name, mono=False, **_ignored):
        self.unit = unit
        self.name = name
        self.mono = bool(mono)
        self.chunkSize = config.standardChunkSize
        self.sampleRate = config.sampleRate

    @property
    def label(self):
        return "%s.%s" % (self.unit.label, self.name.upper())


class Outlet(_Port):
    isOutlet = True

    def __init__(self, unit, name, **opts):
        super().__init__(unit, name, **opts)
        self.connections = []


class Inlet(_Port):
    isInlet = True

    def __init__(self, unit, name, **opts):
        super().__init__(unit, name, **opts)
        self.connected = False
        self.outlet = None
        self.constant = 0
        self.values = [0.0]  # what the reference keeps in the inlet's own chunk: f32 per channel

    def disconnect(self):
        if self.outlet is not None:
            self.outlet.connections.remove(self)
            self.outlet = None
            self.connected = False
            self.values = [0.0] * max(1, len(self.values))

    def setConstant(self, value):  # reference src/Inlet.js:76-93
        if self.outlet is not None:
            self.disconnect()
        self.constant = value
        vals = list(value) if isinstance(value, (list, tuple)) else [value]
        n = max(len(self.values), len(vals))
        self.values = [_f32(vals[c % len(vals)]) for c in range(n)]

    def connect(self, outlet):  # reference src/Inlet.js:40-74
        if getattr(outlet, "isUnit", False):
            outlet = outlet.defaultOutlet
        if self.connected:
            self.disconnect()
        self.connected = True
        self.outlet = outlet
        outlet.connections.append(self)
        a, b = self.unit.circuit, outlet.unit.circuit
        if a is not None and b is not None and a is not b:
            raise RuntimeError("SHIT: Circuit conflict")  # the reference's own message (Inlet.js:58)
        modified = None
        with _deep_recursion():
            if a is not None:
                a.add(outlet.unit)
                modified = a
            elif b is not None:
                b.add(self.unit)
                modified = b
            if modified is not None:
                self.unit.computeProcessIndex()
                outlet.unit.computeProcessIndex()
                modified.computeOrders()


class Event:
    """A host callback at a point in time (reference src/Event.js:3-30): `time` in seconds, `t` the same instant in
    samples.  The callback receives the unit (the reference binds it as `this`); a positive return value
    reschedules it that many seconds later."""

    def __init__(self, time, func, unit=None, circuit=None):
        self.t = time * config.sampleRate
        self.function = func
        self.unit = unit
        self.circuit = circuit

    @property
    def time(self):
        return self.t / config.sampleRate

    def run(self):
        subject = self.unit if self.unit is not None else self.circuit
        again = self.function(subject)
        if isinstance(again, (int, float)) and not isinstance(again, bool) and again > 0:
            return Event(self.time + again, self.function, self.unit, self.circuit)
        return None


def _insert_by_time(events, event):  # after every event that is not later (Circuit.js:49-55)
    for i, other in enumerate(events):
        if event.t < other.t:
            events.insert(i, event)
            return
    events.append(event)


class Unit:
    """reference src/Unit.js"""

    isUnit = True
    _times_used = {}

    def __init__(self):
        object.__setattr__(self, "inlets", {})
        object.__setattr__(self, "outlets", {})
        self.inletsOrdered = []
        self.outletsOrdered = []
        self.events = []
        self.circuit = None
        self.clock = 0
        self.tickInterval = config.standardChunkSize
        self.processIndex = None
        self.nChains = 0
        self.sampleRate = config.sampleRate
        kind = type(self).__name__
        Unit._times_used[kind] = Unit._times_used.get(kind, 0) + 1
        self.label = "%s%d" % (kind, Unit._times_used[kind])

    # `unit.F = 440` / `unit.F = otherUnit` and `unit.OUT`: the accessor is the port name upper-cased
    # (Unit.js:56-67, 82), e.g. inlet "feedbackGain" -> unit.FEEDBACKGAIN
    def _port(self, upper_name, table):
        for name, port in table.items():
            if name.upper() == upper_name:
                return port
        return None

    def __setattr__(self, name, value):
        if name.isupper():
            inlet = self._port(name, self.inlets)
            if inlet is not None:
                if value is None:
                    raise ValueError("Passed bad value to " + inlet.label)
                if isinstance(value, (int, float, list, tuple, np.floating, np.integer)):
                    inlet.setConstant(value)
                elif getattr(value, "isOutlet", False) or getattr(value, "isUnit", False):
                    inlet.connect(value)
                return
        object.__setattr__(self, name, value)

    def __getattr__(self, name):
        if name.isupper():
            port = self._port(name, self.inlets) or self._port(name, self.outlets)
            if port is not None:
                return port
        raise AttributeError(name)

    def addInlet(self, name, **opts):
        inlet = Inlet(self, name, **opts)
        self.inlets[name] = inlet
        self.inletsOrdered.append(inlet)
        return inlet

    def addOutlet(self, name, **opts):
        outlet = Outlet(self, name, **opts)
        self.outlets[name] = outlet
        self.outletsOrdered.append(outlet)
        return outlet

    def chainAfter(self, unit):  # ordering-only edge (Unit.js:88-99)
        if not getattr(unit, "isUnit", False):
            raise TypeError("chainAfter expects a Unit")
        inlet = self.addInlet("chain%d" % self.nChains)
        self.nChains += 1
        outlet = unit.addOutlet("chain%d" % unit.nChains)
        unit.nChains += 1
        inlet.connect(outlet)

    chain = chainAfter

    def chainBefore(self, unit):
        return unit.chainAfter(self)

    @property
    def defaultOutlet(self):
        return self.outletsOrdered[0]

    @property
    def inputUnits(self):
        seen = []
        for inlet in self.inlets.values():
            if inlet.connected and inlet.outlet.unit not in seen:
                seen.append(inlet.outlet.unit)
        return seen

    @property
    def outputUnits(self):
        seen = []
        for outlet in self.outlets.values():
            for inlet in outlet.connections:
                if inlet.unit not in seen:
                    seen.append(inlet.unit)
        return seen

    def computeProcessIndex(self, history=None):
        """1 + the largest index among the inputs not already on the walk; then
        push every dependent whose index is not above ours.  `history` is what
        cuts feedback loops, and WHERE it cuts decides which edge of a loop
        carries the implicit one-chunk delay (reference src/Unit.js:171-209)."""
        history = (history or []) + [self]
        best = -1
        for unit in self.inputUnits:
            if unit in history:
                continue
            if unit.processIndex is None:
                unit.computeProcessIndex(history)
            if unit.processIndex > best:
                best = unit.processIndex
        self.processIndex = best + 1
        for unit in self.outputUnits:
            if unit in history:
                continue
            if unit.processIndex is None or unit.processIndex <= self.processIndex:
                unit.computeProcessIndex(history)
        return self.processIndex

    def getOrBuildCircuit(self):
        return self.circuit if self.circuit is not None else Circuit(self)

    def trigger(self):
        for unit in self.inputUnits:
            unit.trigger()
        return self

    # ---- scheduled events (reference src/UnitOrPatch.js:9-33, src/Unit.js addEvent)
    def addEvent(self, event):
        if self.circuit is not None:
            self.circuit.addEvent(event)
        else:
            _insert_by_time(self.events, event)

    def schedule(self, time, func):
        """Run func(unit) at `time` seconds (a list schedules every entry); it takes effect at the start of the
        chunk that contains it (Circuit.js:23,57-65)."""
        if isinstance(time, (list, tuple)):
            for t in time:
                self.schedule(t, func)
            return None
        self.addEvent(Event(time, func, self))
        return self

    def scheduleTrigger(self, t, val=None):
        self.schedule(t, (lambda unit: unit.trigger()) if val is None else (lambda unit: unit.trigger(val)))


class Circuit:
    """reference src/Circuit.js (construction + ordering only; ticking happens on the GPU)."""

    def __init__(self, *units):
        self.units = []
        self.tickIntervals = []
        self.clock = 0
        self.events = []
        self.promises = []
        with _deep_recursion():
            for unit in units:
                self.add(unit)

    def add(self, unit):  # flood fill over inputs AND outputs (Circuit.js:67-107)
        if unit.circuit is not None and unit.circuit is not self:
            raise RuntimeError("circuit clash, oh god " + unit.label)
        if unit in self.units:
            return None
        self.units.append(unit)
        unit.circuit = self
        if unit.tickInterval not in self.tickIntervals:
            self.tickIntervals = sorted(self.tickIntervals + [unit.tickInterval])
        if unit.events:
            for e in unit.events:
                self.addEvent(e)
        unit.events = None  # from now on the unit's events go straight to the circuit
        for other in unit.inputUnits:
            self.add(other)
        for other in unit.outputUnits:
            self.add(other)
        unit.computeProcessIndex()
        self.computeOrders()
        return True

    def addEvent(self, event):
        event.circuit = self
        _insert_by_time(self.events, event)

    def runEvents(self, beforeT):
        """Run every event due before `beforeT` samples (Circuit.js:57-65)."""
        while self.events and self.events[0].t < beforeT:
            follow_up = self.events.pop(0).run()
            if follow_up is not None:
                self.addEvent(follow_up)

    def computeOrders(self):
        """Stable sort by process index.  While a flood fill is in flight some
        indices are still undefined; the reference's comparator then yields NaN,
        which a JS sort treats as 'equal' (Circuit.js:127-130)."""

        def compare(a, b):
            if a.processIndex is None or b.processIndex is None:
                return 0
            return (a.processIndex > b.processIndex) - (a.processIndex < b.processIndex)

        self.units.sort(key=functools.cmp_to_key(compare))
        self.gcdTickInterval = functools.reduce(math.gcd, self.tickIntervals) if self.tickIntervals else None


# --------------------------------------------------------------------------- units
class Osc(Unit):
    """reference src/components/Osc/Osc.js:7-17"""

    def __init__(self, f=None, waveform=None):
        super().__init__()
        self.addInlet("f", mono=True)
        self.addOutlet("out", mono=True)
        self.F = f or 440
        self.phase = 0
        self.waveform = waveform or "sin"

    @property
    def waveform(self):
        return self._waveform

    @waveform.setter
    def waveform(self, value):
        from .wavetables import WAVEFORMS

        if value not in WAVEFORMS:
            raise ValueError("waveform doesn't exist: %s" % value)
        object.__setattr__(self, "_waveform", value)


class Ramp(Unit):
    """reference src/components/Ramp.js:3-23 — (duration in SAMPLES, y0, y1); idle until trigger()"""

    def __init__(self, duration=None, y0=None, y1=None):
        super().__init__()
        self.addOutlet("out", mono=True)
        self.duration = duration or self.sampleRate
        self.y0 = y0 or 1
        self.y1 = y1 or 0
        self.t = 0
        self.playing = False

    def trigger(self):
        self.playing = True
        self.t = 0
        return self


class Multiply(Unit):
    """reference src/components/Multiply.js:4-12"""

    def __init__(self, a=None, b=None):
        super().__init__()
        self.addInlet("a")
        self.addInlet("b")
        self.addOutlet("out")
        self.A = a or 1
        self.B = b or 1


class Sum(Unit):
    """reference src/components/Sum.js + SignalCombiner.js:3-12"""

    def __init__(self, a=None, b=None):
        super().__init__()
        self.addInlet("a")
        self.addInlet("b")
        self.addOutlet("out")
        self.A = a or 0
        self.B = b or 0

    @staticmethod
    def many(inputs):  # left-deep chain (Sum.js:18-29)
        if len(inputs) == 1:
            return inputs[0]
        acc = Sum(inputs[0], inputs[1])
        for x in inputs[2:]:
            acc = Sum(acc, x)
        return acc


class Filter(Unit):
    """reference src/components/Filter.js:5-22"""

    def __init__(self, input=None, f=None, kind=None):
        super().__init__()
        self.addInlet("in")
        self.addInlet("f", mono=True)
        self.addOutlet("out")
        if input:
            self.IN = input
        if f:
            self.F = f
        self.kind = kind or "LP"
        if self.kind not in ("LP", "HP"):
            raise ValueError("invalid filter type: %s" % self.kind)
        self.lastF = None
        # the kind setter runs the coefficient function with f undefined (Filter.js:63): NaNs,
        # except HP's constant a1 = 0 (Filter.js:80); the first tick overwrites them all
        self.a0 = self.a2 = self.b1 = self.b2 = float("nan")
        self.a1 = 0 if self.kind == "HP" else float("nan")
        self.x1, self.x2, self.y1, self.y2 = [], [], [], []


class Delay(Unit):
    """reference src/components/Delay.js:6-18 — delay in SAMPLES, ring of maxDelay samples"""

    def __init__(self, input=None, delay=None, maxDelay=None):
        super().__init__()
        self.addInlet("in")
        self.addInlet("delay")
        self.addOutlet("out")
        self.maxDelay = maxDelay or self.sampleRate * 5
        self.IN = input or 0
        self.DELAY = delay or 4410


class CircleBuffer:
    """reference src/CircleBuffer.js:3-13"""

    def __init__(self, numberOfChannels=None, lengthInSeconds=None):
        self.numberOfChannels = numberOfChannels or 1
        self.lengthInSeconds = lengthInSeconds
        self.sampleRate = config.sampleRate
        self.lengthInSamples = math.ceil(self.lengthInSeconds * self.sampleRate)


class _CircleBufferNode(Unit):
    """reference src/components/CircleBufferNode.js:7-31"""

    def __init__(self, buffer, offset):
        super().__init__()
        self.t = 0
        self.buffer = buffer
        self.addInlet("offset")
        self.OFFSET = offset or 0


class CircleBufferReader(_CircleBufferNode):
    """reference src/components/CircleBufferReader.js:4-10"""

    def __init__(self, buffer, offset=None):
        super().__init__(buffer, offset)
        self.addOutlet("out")
        self.postWipe = False


class CircleBufferWriter(_CircleBufferNode):
    """reference src/components/CircleBufferWriter.js:4-10"""

    def __init__(self, buffer, offset=None):
        super().__init__(buffer, offset)
        self.addInlet("in")
        self.preWipe = False


class Repeater(Unit):
    """reference src/components/Repeater.js:3-11"""

    def __init__(self, val=None, measuredIn=None):
        super().__init__()
        self.addInlet("in")
        self.addOutlet("out")
        self.measuredIn = measuredIn
        self.IN = val or 0


# --------------------------------------------------------------------------- elementwise maps
class _Binary(Unit):
    _defaults = (0, 0)

    def __init__(self, a=None, b=None):
        super().__init__()
        self.addInlet("a")
        self.addInlet("b")
        self.addOutlet("out")
        self.A = a or self._defaults[0]
        self.B = b or self._defaults[1]


class Subtract(_Binary):
    """reference src/components/Subtract.js:4-11"""
    _defaults = (0, 0)


class Divide(_Binary):
    """reference src/components/Divide.js:3-10"""
    _defaults = (1, 1)


class Pow(Unit):
    """reference src/components/Pow.js:4-11 (no defaults: a missing operand raises, as the reference throws)"""

    def __init__(self, a, b):
        super().__init__()
        self.addInlet("a")
        self.addInlet("b")
        self.addOutlet("out")
        self.A = a
        self.B = b


class _Unary(Unit):
    _default = 0

    def __init__(self, input=None):
        super().__init__()
        self.addInlet("in")
        self.addOutlet("out")
        self.IN = input or self._default


class PolarityInvert(_Unary):
    """reference src/components/PolarityInvert.js:4-9"""


class Abs(_Unary):
    """reference src/components/Abs.js:3-9"""


class DecibelToScaler(_Unary):
    """reference src/components/DecibelToScaler.js:3-8"""


class SemitoneToRatio(_Unary):
    """reference src/components/SemitoneToRatio.js:3-8"""
    _default = 69


class SecondsToSamples(Unit):
    """reference src/components/SecondsToSamples.js:4-8 (no constructor argument)"""

    def __init__(self):
        super().__init__()
        self.addInlet("in")
        self.addOutlet("out")


class FixedMultiply(Unit):
    """reference src/components/FixedMultiply.js:3-9 — (sf, input); sf is a plain number"""

    def __init__(self, sf, input=None):
        super().__init__()
        self.addInlet("in", mono=True)
        self.addOutlet("out", mono=True)
        self.sf = sf
        self.IN = input or 0


class Clip(Unit):
    """reference src/components/Clip.js:4-11 — (threshold) only; `in` stays 0 until set"""

    def __init__(self, threshold):
        super().__init__()
        self.addInlet("in")
        self.addInlet("threshold")
        self.addOutlet("out")
        self.THRESHOLD = threshold


class _HardClip(Unit):
    def __init__(self, input=None, threshold=None):
        super().__init__()
        self.addInlet("in")
        self.addInlet("threshold")
        self.addOutlet("out")
        self.IN = input or 0
        self.THRESHOLD = threshold or 0


class HardClipAbove(_HardClip):
    """reference src/components/HardClipAbove.js:4-12"""


class HardClipBelow(_HardClip):
    """reference src/components/HardClipBelow.js:4-12"""


class Gain(Unit):
    """reference src/components/Gain.js:3-10 — (gain in dB); `in` stays 0 until set"""

    def __init__(self, gain=None):
        super().__init__()
        self.addInlet("in")
        self.addInlet("gain", mono=True)
        self.addOutlet("out")
        self.GAIN = gain or 0


# --------------------------------------------------------------------------- delay / filter family
def _js_round(x):
    return math.floor(x + 0.5)


class FixedDelay(Unit):
    """reference src/components/FixedDelay.js:4-33 — (delayTime in seconds); `in` stays 0 until set"""

    def __init__(self, delayTime):
        super().__init__()
        self.addInlet("in", mono=True)
        self.addOutlet("out", mono=True)
        self.setSeconds(delayTime)
        self.tBuffer = 0

    def setDelayTime(self, tSamples):
        if not tSamples or tSamples < 0.5:
            raise ValueError("Cannot have fixed delay of length 0 samples")  # the reference's message
        self.delayTimeInSamples = _js_round(tSamples)
        self.delayTimeInSeconds = tSamples / self.sampleRate

    def setSeconds(self, duration):
        self.setDelayTime(duration * self.sampleRate)

    def setFrequency(self, f):
        self.setSeconds(1 / f)


class CombFilter(FixedDelay):
    """reference src/components/CombFilter.js:4-9"""

    def __init__(self, delayTime, feedbackGain=None):
        super().__init__(delayTime)
        self.addInlet("feedbackGain", mono=True)
        self.FEEDBACKGAIN = feedbackGain or 0


class AllPass(CombFilter):
    """reference src/components/AllPass.js:4-7"""


class MonoDelay(Unit):
    """reference src/components/MonoDelay.js:3-14 — delay in SAMPLES, fixed 5 s ring"""

    def __init__(self, input=None, delay=None):
        super().__init__()
        self.addInlet("in", mono=True)
        self.addInlet("delay", mono=True)
        self.addOutlet("out", mono=True)
        self.maxDelay = self.sampleRate * 5
        self.IN = input or 0
        self.DELAY = delay or 4410


class ReadBackDelay(Unit):
    """reference src/components/ReadBackDelay.js:4-17"""

    def __init__(self, input=None, delay=None, bufferLength=None):
        super().__init__()
        self.addInlet("in")
        self.addInlet("delay")
        self.addOutlet("out")
        self.bufferLength = bufferLength or config.sampleRate
        self.tBuffer = 0
        self.IN = input or 0
        self.DELAY = delay or 0


class MultiChannelOsc(Unit):
    """reference src/components/Osc/MultiChannelOsc.js:7-17 — one phase per channel of f"""

    def __init__(self, f=None, waveform=None):
        super().__init__()
        self.addInlet("f")
        self.addOutlet("out")
        self.F = f or 440
        self.phase = []
        self.waveform = waveform or "sin"

    @property
    def waveform(self):
        return self._waveform

    @waveform.setter
    def waveform(self, value):
        from .wavetables import WAVEFORMS

        if value not in WAVEFORMS:
            raise ValueError("waveform doesn't exist: %s" % value)
        object.__setattr__(self, "_waveform", value)


# --------------------------------------------------------------------------- rest of the elementwise sweep
class Pan(Unit):
    """reference src/components/Pan.js:3-13 — mono in, stereo out"""

    def __init__(self, input=None, pan=None):
        super().__init__()
        self.addInlet("in", mono=True)
        self.addInlet("pan", mono=True)
        self.addOutlet("out", numberOfChannels=2)
        self.PAN = pan or 0
        self.IN = input or 0
        self.compensationDB = 1.5


class MidiToFrequency(Unit):
    """reference src/components/MidiToFrequency.js:3-9 — the data outlet is called "frequency" """

    def __init__(self, midi=None):
        super().__init__()
        self.addInlet("midi")
        self.addOutlet("frequency")
        self.MIDI = midi or 69


class Rescale(Unit):
    """reference src/components/Rescale.js:3-17 — `in` is not a constructor argument"""

    isRescale = True

    def __init__(self, inLower=None, inUpper=None, outLower=None, outUpper=None):
        super().__init__()
        for name in ("in", "inLower", "inUpper", "outLower", "outUpper"):
            self.addInlet(name)
        self.addOutlet("out")
        self.IN = 0
        self.INLOWER = inLower or -1
        self.INUPPER = inUpper or 1
        self.OUTLOWER = outLower or 0
        self.OUTUPPER = outUpper or 1


class CrossFader(Unit):
    """reference src/components/CrossFader.js:3-14 — dial 0: all A, 1: all B"""

    def __init__(self, a=None, b=None, dial=None):
        super().__init__()
        self.addInlet("a")
        self.addInlet("b")
        self.addInlet("dial", mono=True)
        self.addOutlet("out")
        self.A = a or 0
        self.B = b or 0
        self.DIAL = dial or 0


class VectorMagnitude(Unit):
    """reference src/components/vector/VectorMagnitude.js:5-11 (no constructor argument)"""

    def __init__(self):
        super().__init__()
        self.addInlet("in")
        self.addOutlet("out", mono=True)
        self.IN = [0, 0]


class Timer(Unit):
    """reference src/components/Timer.js:26-32,43-45"""

    def __init__(self):
        super().__init__()
        self.addOutlet("out", mono=True)
        self.t = 0
        self.samplePeriod = 1 / self.sampleRate

    def trigger(self):
        self.t = 0
        return self


class HostSource(Unit):
    """A mono signal the HOST provides (descriptor opcode INPUT): `samples` holds the unit's output from circuit time 0 on
    (zeros past its end).  This is how a unit that only the host can tick enters a device render — the JS host's Noise
    (reference src/components/Noise.js:16-27: Math.random() per sample) is generated this way — and a way to feed recorded
    or externally computed audio into a circuit.  No counterpart class in the reference."""

    def __init__(self, samples):
        super().__init__()
        self.addOutlet("out", mono=True)
        self.samples = np.ascontiguousarray(samples, dtype=np.float32).reshape(-1)


class SampleRateRedux(Unit):
    """reference src/components/SampleRateRedux.js:3-16 — sample & hold every `ammount` samples"""

    def __init__(self, input=None, ammount=None):
        super().__init__()
        self.addInlet("in")
        self.addInlet("ammount", mono=True)
        self.addOutlet("out")
        self.val = [0]
        self.timeSinceLastUpdate = float("inf")
        self.IN = input or 0
        self.AMMOUNT = ammount or 0


class ConcatChannels(Unit):
    """reference src/components/ConcatChannels.js:3-11"""

    def __init__(self, a=None, b=None):
        super().__init__()
        self.addInlet("a")
        self.addInlet("b")
        self.addOutlet("out")
        self.A = a or 0
        self.B = b or 0


class PickChannel(Unit):
    """reference src/components/PickChannel.js:3-11"""

    def __init__(self, input=None, c=None):
        super().__init__()
        self.addInlet("in")
        self.addInlet("c", mono=True)
        self.addOutlet("out", mono=True)
        self.IN = input or 0
        self.C = c or 0


# --------------------------------------------------------------------------- envelopes
class Shape(Unit):
    """reference src/components/Shape/index.js:7-23,107-122 — a table read once over `duration` seconds after trigger()"""

    def __init__(self, shape=None, durationInSeconds=None, min=None, max=None):  # noqa: A002 - the reference's names
        super().__init__()
        self.addInlet("duration", mono=True)
        self.addInlet("min", mono=True)
        self.addInlet("max", mono=True)
        self.addOutlet("out", mono=True)
        self.t = 0
        self.playing = False
        self.finished = False
        self.leftEdge = 0
        self.rightEdge = "shape"
        self.shape = shape or "decay"
        self.DURATION = durationInSeconds or 1
        self.MIN = min or 0
        self.MAX = max or 1

    @property
    def shape(self):
        return self._shape

    @shape.setter
    def shape(self, value):
        from .wavetables import SHAPES

        if value not in SHAPES:
            raise ValueError("%s:\n\tinvalid shape function: %s" % (self.label, value))
        object.__setattr__(self, "_shape", value)

    def trigger(self):
        self.playing = True
        self.t = 0
        return self

    def stop(self):
        self.playing = False


class AHD(Unit):
    """reference src/components/AHD.js:6-34 — attack / hold / decay times in seconds"""

    def __init__(self, attack=None, hold=None, decay=None):
        super().__init__()
        self.addInlet("attack", mono=True)
        self.addInlet("hold", mono=True)
        self.addInlet("decay", mono=True)
        self.addOutlet("out", mono=True)
        self.ATTACK = attack or 0
        self.HOLD = hold or 0
        self.DECAY = decay or 0
        self.state = 0
        self.playing = False
        self.t = 0

    def trigger(self):
        self.state = 1
        self.playing = True
        return self

    def stop(self):
        self.state = 0
        self.playing = False
        return self
