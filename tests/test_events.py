"""Scheduled events through the Python host (dusp_amd/graph.py schedule / scheduleTrigger, render.py's segment loop)
against the reference's `ev_*` vectors: circuits whose host callbacks retrigger envelopes, change inlet constants
or reschedule themselves — including circuits with delay lines, CircleBuffers and feedback edges, whose device memory
has to survive from segment to segment (dusp_program_continue)."""
import json
import os

import numpy as np
import pytest

import cases
import dusp_amd as d
from conftest import GOLDEN, Golden
from dusp_amd import descriptor

with open(os.path.join(GOLDEN, "index_events.json")) as f:
    EVENT_CASES = json.load(f)

USES_DEVICE_TAN = ("ev_filter_sweep", "ev_loop_gain", "ev_loop_delay_change")  # Filter coefficients: device tan()


def test_every_reference_event_case_has_a_python_twin():
    assert sorted(EVENT_CASES) == sorted(cases.event_builders(48000))


@pytest.mark.parametrize("name", EVENT_CASES)
def test_event_graph_extracts_to_the_reference_descriptor(name):
    """Before the first tick the Python mirror must hold what the reference's objects held (events pending)."""
    g = Golden(name)
    ex = descriptor.extract(cases.build_event_case(name, g.sample_rate), allow_events=True)
    assert ex.words.size == g.desc.size and ((ex.words == g.desc) | (np.isnan(ex.words) & np.isnan(g.desc))).all()
    assert ex.circuit.events, "the case has no pending event"
    with pytest.raises(descriptor.DuspError, match="scheduled events"):
        descriptor.extract(cases.build_event_case(name, g.sample_rate))


def test_event_order_and_rescheduling_follow_the_reference():
    d.configure(48000)
    log = []
    osc = d.Osc(100)
    osc.schedule(0.02, lambda u: log.append(("b", u.label)))
    osc.schedule(0.01, lambda u: log.append(("a", u.label)))
    osc.schedule(0.02, lambda u: log.append(("c", u.label)))  # same instant: after the one already queued

    def repeat(u):
        log.append(("r", round(u.circuit.events[0].t) if u.circuit.events else None))
        return 0.015 if len([x for x in log if x[0] == "r"]) < 3 else None
    osc.schedule(0.005, repeat)
    circuit = osc.getOrBuildCircuit()
    assert [e.t for e in circuit.events] == [240.0, 480.0, 960.0, 960.0]
    circuit.runEvents(256)                      # events with t < 256 only (Circuit.js:57-65)
    assert [x[0] for x in log] == ["r"] and len(circuit.events) == 4   # rescheduled at 0.005 + 0.015 = 0.02 s
    circuit.runEvents(48000)
    assert [x[0] for x in log] == ["r", "a", "b", "c", "r", "r"]
    assert not circuit.events


@pytest.mark.gpu
@pytest.mark.parametrize("name", EVENT_CASES)
def test_python_host_renders_event_cases_like_the_reference(name):
    g = Golden(name)
    target = cases.build_event_case(name, g.sample_rate)
    cd = d.renderChannelData(target, g.meta["duration"])
    assert len(cd) == g.n_channels and cd.sampleRate == g.sample_rate and cd[0].size == g.n_samples
    got = g.windowed(np.stack(cd))
    if name in USES_DEVICE_TAN:
        scale = float(np.max(np.abs(g.pcm)))
        assert float(np.max(np.abs(got.astype(np.float64) - g.pcm))) <= 1e-5 * scale
    else:
        assert np.array_equal(got, g.pcm), "first mismatch at %d" % int(np.argmax(got != g.pcm))
    circuit = target.circuit
    assert circuit.clock == -(-g.n_samples // 256) * 256 and not [e for e in circuit.events if e.t < circuit.clock - 256]


@pytest.mark.gpu
def test_state_is_written_back_into_the_python_objects():
    d.configure(48000)
    osc = d.Osc(440.5)
    filt = d.Filter(osc, 1200)
    d.renderChannelData(filt, 1000 / 48000)
    assert osc.phase == (1024 * 440.5) % 48000 and filt.circuit.clock == 1024
    assert filt.lastF == 1200 and len(filt.y1) == 1 and filt.y1[0] != 0
    with pytest.raises(descriptor.DuspError, match="already been ticked"):
        d.renderChannelData(filt, 0.01)
