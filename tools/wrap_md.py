"""Re-flow the prose of a markdown file to a column limit (default 120): paragraphs and list items are re-wrapped, tables, headings, fenced code and
lines that are one unbreakable token are left alone.   usage: python tools/wrap_md.py FILE [WIDTH]"""
import re
import sys
import textwrap


def wrap(path, width=120):
    out, para, fence = [], [], False
    lines = open(path).read().split("\n")

    def flush():
        if not para:
            return
        first = para[0]
        m = re.match(r"^(\s*(?:[*\-+]|\d+\.)\s+)", first)
        indent = m.group(1) if m else re.match(r"^(\s*)", first).group(1)
        body = " ".join([first[len(indent):].strip()] + [p.strip() for p in para[1:]])
        sub = " " * len(indent)
        out.extend(textwrap.wrap(body, width=width, initial_indent=indent, subsequent_indent=sub, break_long_words=False, break_on_hyphens=False) or [indent.rstrip()])
        del para[:]
    for ln in lines:
        if ln.strip().startswith("```"):
            flush()
            fence = not fence
            out.append(ln)
            continue
        if fence or ln.startswith("|") or ln.startswith("#") or not ln.strip():
            flush()
            out.append(ln)
            continue
        if re.match(r"^\s*(?:[*\-+]|\d+\.)\s+", ln) and para:       # a new list item ends the previous paragraph
            flush()
        para.append(ln)
    flush()
    open(path, "w").write("\n".join(out))


if __name__ == "__main__":
    wrap(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 120)
