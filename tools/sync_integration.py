#!/usr/bin/env python3
"""here: copies the `//[doc:name]` blocks of oracle/bind_check.cpp (the binding that is compiled against the reference's headers and
linked with its objects) into INTEGRATION.md between `<!-- bind:name -->` and `<!-- /bind -->`, so the document shows the code that
was built, not a paraphrase (tests/test_binding.py checks it)."""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_binding import doc_blocks  # noqa: E402

blocks = doc_blocks()
p = os.path.join(ROOT, "INTEGRATION.md")
doc = open(p).read()


def put(m):
    name = m.group(1)
    return "<!-- bind:%s -->\n```cpp\n%s\n```\n<!-- /bind -->" % (name, blocks[name].strip("\n"))


new, n = re.subn(r"<!-- bind:([a-z_]+) -->.*?<!-- /bind -->", put, doc, flags=re.S)
open(p, "w").write(new)
print("INTEGRATION.md: %d blocks synchronised (%s)" % (n, ", ".join(sorted(blocks))))
