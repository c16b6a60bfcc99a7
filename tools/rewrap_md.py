#!/usr/bin/env python3
"""Re-wrap a Markdown file at <= WIDTH columns (default 160) so that it can be read and diffed: paragraphs and list items are re-flowed with their indentation and bullet
kept, code fences and headings are left alone, and a table with a row wider than WIDTH is turned into a list (one item per row, `header: cell` pairs) because Markdown
cannot wrap a table row.  usage: tools/rewrap_md.py in.md out.md [width]"""
import re
import sys
import textwrap


def flush(par, out, width):
    if not par:
        return
    first = par[0]
    m = re.match(r"^(\s*)((?:[-*+]|\d+\.|\(\w+\))\s+)?", first)
    indent, bullet = m.group(1), m.group(2) or ""
    text = " ".join([first[len(indent) + len(bullet):].strip()] + [ln.strip() for ln in par[1:]])
    sub = indent + " " * len(bullet)
    out.extend(textwrap.wrap(text, width=width, initial_indent=indent + bullet, subsequent_indent=sub, break_long_words=False, break_on_hyphens=False) or [indent + bullet])
    par.clear()


def table_to_list(rows, out, width):
    cells = [[c.strip() for c in r.strip().strip("|").split("|")] for r in rows]
    header, body = cells[0], [c for c in cells[2:]]
    for c in body:
        parts = []
        for h, v in zip(header, c):
            if v:
                parts.append(("%s: %s" % (h, v)) if h else v)
        par = ["* " + "; ".join(parts)]
        flush(par, out, width)


def main():
    src, dst = sys.argv[1], sys.argv[2]
    width = int(sys.argv[3]) if len(sys.argv) > 3 else 160
    lines = open(src).read().split("\n")
    out, par, i, fence = [], [], 0, False
    while i < len(lines):
        ln = lines[i]
        if ln.lstrip().startswith("```"):
            flush(par, out, width); fence = not fence; out.append(ln); i += 1; continue
        if fence:
            out.append(ln); i += 1; continue
        if ln.lstrip().startswith("|"):
            flush(par, out, width)
            rows = []
            while i < len(lines) and lines[i].lstrip().startswith("|"):
                rows.append(lines[i]); i += 1
            if max(len(r) for r in rows) <= width or len(rows) < 3:
                out.extend(rows)
            else:
                table_to_list(rows, out, width)
            continue
        if not ln.strip() or ln.startswith("#"):
            flush(par, out, width); out.append(ln); i += 1; continue
        starts_item = re.match(r"^\s*(?:[-*+]|\d+\.)\s+", ln) is not None
        if par and (starts_item or (len(ln) - len(ln.lstrip())) < (len(par[0]) - len(par[0].lstrip()))):
            flush(par, out, width)
        par.append(ln); i += 1
    flush(par, out, width)
    open(dst, "w").write("\n".join(out))
    print("%s: %d lines, longest %d columns" % (dst, len(out), max(len(x) for x in out)))


if __name__ == "__main__":
    main()
