"""The JavaScript host side (dusp_amd/js): graph mirror + extractor + N-API addon.

Node.js is the reference's own host language; these tests drive the JS surface with `node` and read
back one JSON line.  CPU tests cover the mirror and the addon's loading / error behaviour; the GPU
test renders every golden case through `renderChannelData(unit, duration)`.
"""
import json
import os
import shutil
import subprocess

import pytest

from conftest import ROOT

NODE = shutil.which("node")
needs_node = pytest.mark.skipif(NODE is None, reason="node is not installed")
ADDON = os.path.join(ROOT, "dusp_amd", "js", "addon", "dusp_napi.node")
GOLDEN = os.path.join(ROOT, "tests", "golden")


def run_js(script, *args, ok_codes=(0,)):
    if not os.path.exists(ADDON):
        subprocess.check_call(["make", "-C", os.path.dirname(ADDON), "-s"])
    p = subprocess.run([NODE, os.path.join(ROOT, "tests", "js", script)] + list(args), cwd=ROOT,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    lines = [ln for ln in p.stdout.decode().splitlines() if ln.startswith("{")]
    assert p.returncode in ok_codes and lines, "exit %d\n%s\n%s" % (p.returncode, p.stdout.decode()[-2000:], p.stderr.decode()[-2000:])
    return json.loads(lines[-1])


@needs_node
def golden_count(sr):
    """event-free + event cases generated from the reference at this sample rate (tests/golden/index*.json)"""
    sfx = "" if sr == 48000 else "_sr%d" % sr
    n = 0
    for stem in ("index", "index_events"):
        f = os.path.join(GOLDEN, stem + sfx + ".json")
        if os.path.exists(f):
            n += len(json.load(open(f)))
    return n


@needs_node
@pytest.mark.parametrize("sr", [48000, 44100])
def test_js_graph_mirror_extracts_the_reference_descriptors(sr):
    count = golden_count(sr)
    assert count >= (101 if sr == 48000 else 5)
    rep = run_js("check_descriptors.js", "--sampleRate=%d" % sr)
    assert rep["sampleRate"] == sr and rep["checked"] == count and rep["bad"] == 0 and rep["unifyOk"]


@needs_node
def test_addon_loads_and_fails_loudly_without_gpu():
    rep = run_js("check_addon.js", "--sampleRate=48000")
    assert rep["abi"] == 2 and "gfx950" in rep["version"]
    assert set(rep["exports"]) >= {"ctxCreate", "tableUpload", "programBuild", "programContinue", "render", "stateDownload"}
    assert rep["nullRejects"] == "renderAudioBuffer expects an outlet"  # the reference's own string
    if not rep["gpu"]:
        assert rep["ctxErrorIsString"] is True
        assert rep["renderError"].startswith("dusp-hip: no usable HIP device")  # rejected, no CPU fallback


@needs_node
@pytest.mark.gpu
@pytest.mark.parametrize("sr", [48000, 44100])
def test_js_render_channel_data_matches_reference_golden(sr):
    count = golden_count(sr)
    rep = run_js("check_render.js", "--sampleRate=%d" % sr)
    assert rep.get("fatal") is None, rep
    assert rep["checked"] == count and not rep["failed"], rep["failed"]
    assert rep["exact"] + rep["withinTol"] == count
    assert rep["oscPhaseAfter"] == rep["oscPhaseExpected"] and rep["clockAfter"] == 1024
    assert "already been ticked" in rep["secondRenderRejects"]
    assert rep["manyMatchesSolo"] is True
    assert rep["unsupported"].startswith("dusp-hip: unit type not supported")
