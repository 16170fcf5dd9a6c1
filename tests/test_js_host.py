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


# Vectors whose graphs the reference built with its patches (src/patches/*, host-side builders that this package does not
# mirror): the event-free ones are rendered from the stored descriptor (extracted from the reference's live objects); the
# ones that need the patch OBJECTS on the host (scheduled events on a patch, a patch built around Noise) are not run here.
NEEDS_PATCH_OBJECTS = ()


def golden_names_js(sr):
    """event-free + event cases generated from the reference at this sample rate (tests/golden/index*.json)"""
    sfx = "" if sr == 48000 else "_sr%d" % sr
    names = []
    for stem in ("index", "index_events", "index_host"):
        f = os.path.join(GOLDEN, stem + sfx + ".json")
        if os.path.exists(f):
            names += json.load(open(f))
    return [n for n in names if not n.startswith(NEEDS_PATCH_OBJECTS)]


def golden_count(sr, built_here_only=False):
    names = golden_names_js(sr)
    return len([n for n in names if not n.startswith("patch_")]) if built_here_only else len(names)


@needs_node
@pytest.mark.parametrize("sr", [48000, 44100])
def test_js_graph_mirror_extracts_the_reference_descriptors(sr):
    count = golden_count(sr, built_here_only=True)
    assert count >= (101 if sr == 48000 else 5)
    rep = run_js("check_descriptors.js", "--sampleRate=%d" % sr)
    assert rep["sampleRate"] == sr and rep["checked"] == count and rep["bad"] == 0 and rep["unifyOk"]
    # lib/dusp.js prints every case's graph exactly as the reference's dusp() did (labels renumbered), and most of those
    # strings are fixed points of unDusp -> dusp
    assert rep["strings"] == count and rep["badStrings"] == 0 and rep["roundTrips"] >= 0.75 * count


@needs_node
def test_string_front_end_matches_the_reference_parser_and_constructors():
    """dusp_amd/js/lib/parse.js + unDusp.js against trees / descriptors captured from the reference's own parser
    (the copy inside its browserify bundle; oracle/js/gen_golden_strings.js): 800+ strings incl. two fuzzers."""
    rep = run_js("check_strings.js", "--sampleRate=48000")
    assert rep["trees"] >= 800 and not rep["treeMismatches"], rep["treeMismatches"][:3]
    assert rep["graphs"] >= 31 and not rep["graphMismatches"], rep["graphMismatches"]
    assert rep["rejected"] == 3


@needs_node
@pytest.mark.gpu
def test_strings_render_like_the_reference():
    rep = run_js("check_strings.js", "--sampleRate=48000", "--render")
    assert rep.get("fatal") is None, rep
    assert not rep["renderFailures"] and not rep["graphMismatches"], rep
    assert rep["rendered"] == rep["graphs"] - rep["rejected"] >= 28


@needs_node
@pytest.mark.gpu
def test_render_stream_frames_and_wav():
    """RenderStream (interleaved f32 LE frames, auto-normalising, block-wise continuation of one device program)
    against frames captured from the reference's RenderStream; WAV encode/decode; device-side interleave."""
    rep = run_js("check_stream.js", "--sampleRate=48000")
    assert rep.get("fatal") is None, rep
    assert rep["checked"] == 10 and not rep["failed"], rep["failed"]
    assert rep["wavFloatExact"] and rep["wavPcm16Close"] and rep["deviceFramesMatch"]


@needs_node
def test_wav_header_layout():
    out = subprocess.run([NODE, "-e", """
const {encodeWav, decodeWav} = require('./dusp_amd/js/lib/wav')
const cd = [Float32Array.from([0, 0.5, -1, 1.5]), Float32Array.from([1, -0.25, 0, -2])]; cd.sampleRate = 48000
const f = encodeWav(cd), p = encodeWav(cd, {bitDepth: 16})
console.log(JSON.stringify({flen: f.length, plen: p.length, riff: f.toString('ascii', 0, 4) + f.toString('ascii', 8, 16), ftag: f.readUInt16LE(20),
  ptag: p.readUInt16LE(20), rate: f.readUInt32LE(24), align: f.readUInt16LE(32), frames: Array.from(decodeWav(f).channelData[1]),
  pcm: [p.readInt16LE(44), p.readInt16LE(46), p.readInt16LE(52), p.readInt16LE(56), p.readInt16LE(58)]}))
"""], cwd=ROOT, stdout=subprocess.PIPE, check=True).stdout
    rep = json.loads(out)
    assert rep["riff"] == "RIFFWAVEfmt " and rep["ftag"] == 3 and rep["ptag"] == 1 and rep["rate"] == 48000 and rep["align"] == 8
    assert rep["flen"] == 12 + 8 + 18 + 12 + 8 + 32 and rep["plen"] == 44 + 16
    assert rep["frames"] == [1, -0.25, 0, -2]
    assert rep["pcm"] == [0, 32767, -32767, 32767, -32767]  # clamped to full scale


@needs_node
def test_addon_loads_and_fails_loudly_without_gpu():
    rep = run_js("check_addon.js", "--sampleRate=48000")
    assert rep["abi"] == 7 and "gfx950" in rep["version"]
    assert set(rep["exports"]) >= {"ctxCreate", "tableUpload", "programBuild", "programContinue", "render", "stateDownload", "deviceCount"}
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
    assert rep["fromDescriptor"] == count - golden_count(sr, built_here_only=True)  # the patch_* vectors: descriptor in, PCM out
    assert rep["oscPhaseAfter"] == rep["oscPhaseExpected"] and rep["clockAfter"] == 1024
    assert "already been ticked" in rep["secondRenderRejects"]
    assert rep["manyMatchesSolo"] is True
    assert rep["manyRetriggered"] is True
    # renderMany sharded over devices (here: ONE card listed several times — two / three contexts side by side), bit for bit the one-context render
    assert rep["devices"] >= 1 and rep["shardedTwice"] is True and rep["shardedThrice"] is True and rep["shardedDefault"] is True
    assert rep["shardedMoreDevicesThanVoices"] is True and rep["shardedLoops"] is True
    assert rep["badDevice"].startswith("dusp-hip: renderMany: device 99 is not one of the")
    from dusp_amd.shard import instance_range
    assert [tuple(x) for x in rep["instanceRange"]] == [instance_range(*a) for a in ((10, 0, 3), (10, 1, 3), (10, 2, 3), (2, 4, 5), (65536, 7, 8))]
    assert rep["manyRefusesHostTicked"].startswith("dusp-hip: renderMany does not take circuits with host-ticked units")
    assert rep["unsupported"].startswith("dusp-hip: unit type not supported")
    assert rep["thenWithDeviceMemory"].startswith("dusp-hip: the circuit was rewired during the render (the circuit holds delay lines")
