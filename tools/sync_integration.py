#!/usr/bin/env python3
"""Regenerate the verbatim listing of the shipped plug-in header inside INTEGRATION.md (between the BEGIN/END markers).
`python tools/sync_integration.py --check` exits 1 when the document is stale (used by tests/test_boundary_surface.py)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DOC = os.path.join(ROOT, "INTEGRATION.md")
HDR = os.path.join(ROOT, "pam_amd", "csrc", "host", "dynamics", "awfl_amd", "Dycore.h")
BEGIN, END = "<!-- BEGIN Dycore.h -->", "<!-- END Dycore.h -->"


def render():
    doc = open(DOC).read()
    a, b = doc.index(BEGIN) + len(BEGIN), doc.index(END)
    return doc[:a] + "\n```cpp\n" + open(HDR).read().rstrip("\n") + "\n```\n" + doc[b:]


if __name__ == "__main__":
    new = render()
    if "--check" in sys.argv:
        sys.exit(0 if new == open(DOC).read() else 1)
    open(DOC, "w").write(new)
