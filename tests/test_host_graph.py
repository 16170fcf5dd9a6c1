"""Host mirror (dusp_amd/graph.py + descriptor.py): same descriptor as the reference objects gave."""
import numpy as np
import pytest

import cases
import dusp_amd as d
from conftest import ALL_GOLDEN, Golden
from dusp_amd import descriptor


# patch_* vectors come from the reference's patch builders, which only the JS host mirrors (dusp_amd/js/lib/patches.js);
# their descriptors still feed the oracle and GPU parity tests
@pytest.mark.parametrize("name", [n for n in ALL_GOLDEN if not n.startswith("patch_")])
def test_python_graph_extracts_to_the_reference_descriptor(name):
    g = Golden(name)
    ex = descriptor.extract(cases.build(name, g.sample_rate))
    # identical words = identical constants, state, ring table AND unit order (the reference's
    # computeProcessIndex + stable sort, restated in graph.py)
    assert ex.words.size == g.desc.size
    same = (ex.words == g.desc) | (np.isnan(ex.words) & np.isnan(g.desc))
    assert same.all(), "first differing word: %d" % int(np.argmin(same))


@pytest.mark.parametrize("sr,fname", [(48000, "wavetables.json"), (44100, "wavetables_sr44100.json")])
def test_host_tables_are_the_references(sr, fname):
    """The tables the Python host uploads (wave tables 0-4, Shape tables 5-8) hash to what the reference held."""
    import hashlib
    import json
    import os

    from conftest import GOLDEN
    from dusp_amd.wavetables import N_TABLES, TABLE_NAMES, make_table
    with open(os.path.join(GOLDEN, fname)) as f:
        meta = json.load(f)["tables"]
    assert len(TABLE_NAMES) == N_TABLES == 9
    for tid, w in enumerate(TABLE_NAMES):
        t = make_table(tid, sr)
        assert t.dtype == np.float32 and t.size == meta[w]["length"]
        assert hashlib.sha256(t.tobytes()).hexdigest() == meta[w]["sha256"], w


def test_feedback_loop_order_is_the_references():
    # observed on the reference for the G6 loop
    f = cases.build("loop_220", 48000)
    order = ["%s:%d" % (type(u).__name__, u.processIndex) for u in f.circuit.units] if f.circuit else None
    ex = descriptor.extract(f)
    order = ["%s:%d" % (type(u).__name__, u.processIndex) for u in ex.circuit.units]
    assert order == ["Osc:0", "Multiply:0", "Sum:3", "Delay:4", "Filter:5"]


def test_constructor_defaults_follow_the_or_idiom():
    d.configure(48000)
    assert d.Osc(0).F.values == [440.0]            # `f || 440` (Osc.js:14)
    assert d.Multiply(d.Osc(1), 0).B.values == [1.0]  # `b || 1` (Multiply.js:11)
    r = d.Ramp(0, 0, 0)
    assert (r.duration, r.y0, r.y1, r.playing) == (48000, 1, 0, False)  # Ramp.js:8-10,13
    m = d.Multiply(d.Osc(1), 2)
    m.B = 0
    assert m.B.values == [0.0]                     # setting the inlet afterwards does give 0
    assert d.Osc(0.1).F.values == [float(np.float32(0.1))]  # constants are f32-rounded (Inlet.js:88-91)
    assert d.Delay().maxDelay == 48000 * 5 and d.Delay().DELAY.values == [4410.0]


def test_unify_turns_differing_constants_into_params():
    d.configure(48000)
    voices = [d.Multiply(d.Osc(10 * k), d.Ramp(48000, 1, 0).trigger()) for k in range(1, 9)]
    uni = descriptor.unify([descriptor.extract(v) for v in voices])
    assert uni.n_params == 1 and uni.n_instances == 8
    assert uni.params.shape == (1, 8) and list(uni.params[0]) == [10.0 * k for k in range(1, 9)]
    assert uni.words[6] == 1
    with pytest.raises(descriptor.DuspError):
        descriptor.unify([descriptor.extract(d.Osc(1)), descriptor.extract(d.Ramp())])


def test_rejections_use_the_reference_messages():
    with pytest.raises(descriptor.DuspError, match="renderAudioBuffer expects an outlet"):
        descriptor.extract(None)
    with pytest.raises(descriptor.DuspError, match="renderAudioBuffer expects an outlet"):
        descriptor.extract(object())
