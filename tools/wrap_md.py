"""Re-wrap a markdown file at <= WIDTH columns: paragraphs and list items are wrapped (continuation lines keep the item's indent), code
fences are left alone, table rows that do not fit become list items (first cell bold, the other cells after em-dashes).
    python tools/wrap_md.py <in.md> <out.md> [width]"""
import re
import sys
import textwrap

src, dst = sys.argv[1], sys.argv[2]
WIDTH = int(sys.argv[3]) if len(sys.argv) > 3 else 160
out, fence = [], False
lines = open(src, encoding="utf-8").read().split("\n")
i = 0
while i < len(lines):
    line = lines[i]
    if line.lstrip().startswith("```"):
        fence = not fence
        out.append(line); i += 1; continue
    if fence:
        out.append(line); i += 1; continue
    if line.startswith("|"):
        j = i
        while j < len(lines) and lines[j].startswith("|"):
            j += 1
        block = lines[i:j]
        if all(len(b) <= WIDTH for b in block):
            out.extend(block); i = j; continue
        for n, row in enumerate(block):           # a table with a row that does not fit: every row becomes a list item
            cells = [c.strip() for c in row.strip().strip("|").split("|")]
            if all(re.fullmatch(r":?-+:?", c) for c in cells):
                continue
            if n == 0:
                out.extend(textwrap.wrap("*(" + " — ".join(cells) + ")*", WIDTH, break_long_words=False, break_on_hyphens=False)); out.append(""); continue
            text = "* **%s**" % cells[0] + "".join(" — " + c for c in cells[1:] if c)
            out.extend(textwrap.wrap(text, WIDTH, subsequent_indent="  ", break_long_words=False, break_on_hyphens=False))
        i = j; continue
    if len(line) <= WIDTH:
        out.append(line); i += 1; continue
    m = re.match(r"^(\s*)([*+-] |\d+\. |> )?", line)
    indent = m.group(1) + (" " * len(m.group(2)) if m.group(2) else "")
    out.extend(textwrap.wrap(line, WIDTH, initial_indent="", subsequent_indent=indent, break_long_words=False, break_on_hyphens=False))
    i += 1
open(dst, "w", encoding="utf-8").write("\n".join(out))
print("lines %d -> %d, longest %d" % (len(lines), len(out), max(len(l) for l in out)))
