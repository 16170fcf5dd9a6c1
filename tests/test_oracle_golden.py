"""The CPU oracle (oracle/dusp_oracle.c) against every vector the JS reference produced.

This is what pins the oracle: each golden file was written by the reference's own
renderChannelData (oracle/js/gen_golden.js); the oracle must reproduce the stored
windows bit for bit AND the sha256 of the full-length PCM."""
import hashlib
import json
import os

import numpy as np
import pytest

from conftest import ALL_GOLDEN, GOLDEN, Golden


@pytest.mark.parametrize("name", ALL_GOLDEN)
def test_oracle_matches_reference_bit_for_bit(oracle, name):
    g = Golden(name)
    pcm = oracle.render(g.desc, g.n_samples)
    assert pcm.shape == (g.n_channels, g.n_samples)
    assert g.windowed(pcm).tobytes() == g.pcm.tobytes()
    assert hashlib.sha256(pcm.tobytes()).hexdigest() == g.meta["sha256_full"]


@pytest.mark.parametrize("sr,fname", [(48000, "wavetables.json"), (44100, "wavetables_sr44100.json")])
def test_oracle_wavetables_match_reference(oracle, sr, fname):
    with open(os.path.join(GOLDEN, fname)) as f:
        meta = json.load(f)
    for tid, w in enumerate(["sin", "saw", "square", "triangle", "8bit", "decay", "attack", "semiSine", "decaySquared"]):
        t = oracle.wavetable(tid, sr)
        assert t.size == meta["tables"][w]["length"] == sr + 1
        assert hashlib.sha256(t.tobytes()).hexdigest() == meta["tables"][w]["sha256"], w


def test_documented_reference_values(oracle):
    # values observed on the reference itself
    t = oracle.wavetable(0, 48000)
    assert float(t[1]) == 1.3089696585666388e-4 and float(t[48000]) == -1.3089696585666388e-4
    g = Golden("osc440_1s")
    pcm = oracle.render(g.desc, 48000)[0]
    assert [float(x) for x in pcm[:4]] == [0.057562828063964844, 0.1149347648024559, 0.17192555963993073,
                                           0.22834619879722595]
    assert pcm[47999] == 0


def test_oracle_rejects_garbage(oracle):
    with pytest.raises(oracle.OracleError):
        oracle.render(np.zeros(12), 16)
