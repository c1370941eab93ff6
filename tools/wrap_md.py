#!/usr/bin/env python3
"""Re-flow the prose of a Markdown file at a column (default 140) without touching tables, headings, fenced or indented (>= 4 spaces) blocks.
Consecutive prose lines of one paragraph or list item are joined and wrapped again; continuation lines take the item text's indentation.
The text itself never changes (whitespace-normalised, the file reads the same before and after)."""
import re
import sys

ITEM = re.compile(r"^(\s*)((?:[*+-]|\d+\.)\s+)(.*)$")


def is_prose(ln):
    return bool(ln.strip()) and not (ln.startswith("|") or ln.startswith("#") or ln.startswith("    ") or ln.startswith("\t") or ln.lstrip().startswith("```"))


def wrap(first, cont, text, width):
    out, cur = [], first
    for word in text.split():
        if len(cur) + (1 if cur.strip() else 0) + len(word) > width and cur.strip() and cur != first.rstrip() + "":
            out.append(cur.rstrip())
            cur = cont + word
        else:
            cur = cur + word if (not cur.strip() or cur.endswith(" ")) else cur + " " + word
    out.append(cur.rstrip())
    return out


def main():
    path = sys.argv[1]
    width = int(sys.argv[2]) if len(sys.argv) > 2 else 140
    lines = open(path).read().split("\n")
    res, fenced, k = [], False, 0
    while k < len(lines):
        ln = lines[k]
        if ln.lstrip().startswith("```"):
            fenced = not fenced
            res.append(ln)
            k += 1
            continue
        if fenced or not is_prose(ln):
            res.append(ln)
            k += 1
            continue
        m = ITEM.match(ln)
        if m:
            first, cont, text = m.group(1) + m.group(2), m.group(1) + " " * len(m.group(2)), m.group(3)
        else:
            ind = re.match(r"^(\s*)", ln).group(1)
            first, cont, text = ind, ind, ln.strip()
        k += 1
        while k < len(lines) and is_prose(lines[k]) and not ITEM.match(lines[k]) and not lines[k - 1].endswith("  "):
            text += " " + lines[k].strip()
            k += 1
        res.extend(wrap(first, cont, text, width))
    open(path, "w").write("\n".join(res))


if __name__ == "__main__":
    main()
