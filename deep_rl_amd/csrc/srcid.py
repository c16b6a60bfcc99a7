#!/usr/bin/env python3
"""mi_source_id: 12 hex digits of the sha256 over the CODE of the files named on the command line, in that order — comments and white space are stripped first, so that
an edit of a comment (the header's documentation, a note in a kernel) does not change the identity of the binary while any change of a token does.  Used by the Makefile."""
import hashlib
import re
import sys


def code_only(text):
    # string literals are kept as they are; // and /* */ comments go; then every run of white space collapses
    out, i, n = [], 0, len(text)
    while i < n:
        c = text[i]
        if c == '"' or c == "'":
            j = i + 1
            while j < n and text[j] != c:
                j += 2 if text[j] == "\\" else 1
            out.append(text[i:j + 1]); i = j + 1
        elif text.startswith("//", i):
            j = text.find("\n", i)
            i = n if j < 0 else j
        elif text.startswith("/*", i):
            j = text.find("*/", i + 2)
            i = n if j < 0 else j + 2
            out.append(" ")
        else:
            out.append(c); i += 1
    return re.sub(r"\s+", " ", "".join(out)).strip()


h = hashlib.sha256()
for path in sys.argv[1:]:
    h.update(code_only(open(path, encoding="utf-8", errors="replace").read()).encode())
    h.update(b"\0")
print(h.hexdigest()[:12])
