#!/usr/bin/env python3
"""Wrap the prose lines of a Markdown file at a column (default 140) without touching tables, headings, fenced or indented code.
A long line is broken at spaces; continuation lines take the line's own indentation (list items: the text's)."""
import re
import sys


def wrap_line(line, width):
    m = re.match(r"^(\s*)((?:[*+-]|\d+\.)\s+)?(.*)$", line)
    indent, bullet, text = m.group(1), m.group(2) or "", m.group(3)
    first = indent + bullet
    cont = indent + " " * len(bullet)
    out, cur = [], first
    for word in text.split(" "):
        if cur.strip() and len(cur) + 1 + len(word) > width and len(cur) > len(cont):
            out.append(cur.rstrip())
            cur = cont + word
        else:
            cur = (cur + " " + word) if (cur.strip() and not cur.endswith(" ")) or (cur.strip() and cur != first) else cur + word
    out.append(cur.rstrip())
    return out


def main():
    path = sys.argv[1]
    width = int(sys.argv[2]) if len(sys.argv) > 2 else 140
    lines = open(path).read().split("\n")
    res, fenced = [], False
    for ln in lines:
        if ln.lstrip().startswith("```"):
            fenced = not fenced
            res.append(ln)
            continue
        if fenced or len(ln) <= width or ln.startswith("|") or ln.startswith("#") or ln.startswith("    ") or ln.startswith("\t"):
            res.append(ln)
            continue
        res.extend(wrap_line(ln, width))
    open(path, "w").write("\n".join(res))


if __name__ == "__main__":
    main()
