"""The opcode table lives in four places — dusp_amd/js/lib/ops.js, dusp_amd/descriptor.py, dusp_amd/csrc/device_types.hpp,
oracle/dusp_oracle.c — this keeps them in step."""
import os
import re

from conftest import ROOT
from dusp_amd import descriptor


def _enum_names(text, first):
    body = text[text.index(first):]
    body = body[:body.index("}")]
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    body = re.sub(r"//[^\n]*", "", body)
    names, value = {}, 0
    for item in body.split(","):
        item = item.strip()
        if not item:
            continue
        m = re.match(r"(OP_[A-Z0-9_]+)\s*(?:=\s*(\w+))?$", item)
        assert m, item
        if m.group(2) is not None:
            if not m.group(2).isdigit():
                continue  # aliases such as OP_MAP_FIRST = OP_SUBTRACT
            value = int(m.group(2))
        else:
            value += 1
        names[m.group(1)] = value
    return names


def test_opcode_tables_agree():
    js = open(os.path.join(ROOT, "dusp_amd", "js", "lib", "ops.js")).read()
    body = js[js.index("const OP = Object.freeze({"):]
    body = re.sub(r"//[^\n]*", "", body[:body.index("})")])
    js_ops = {"OP_" + k: int(v) for k, v in re.findall(r"([A-Z0-9_]+):\s*(\d+)", body)}
    py_ops = {k: v for k, v in vars(descriptor).items() if k.startswith("OP_") and isinstance(v, int)}
    hpp = _enum_names(open(os.path.join(ROOT, "dusp_amd", "csrc", "device_types.hpp")).read(), "OP_OSC = 1")
    c = _enum_names(open(os.path.join(ROOT, "oracle", "dusp_oracle.c")).read(), "OP_OSC = 1")
    assert len(js_ops) >= 42
    assert js_ops == py_ops == hpp == c
