#!/usr/bin/env python3
"""Dev tool: re-wrap the paragraphs and list items of a markdown file at 120 columns (tables, headings and code blocks are left
alone).  usage: python tools/wrap_md.py DESIGN.md"""
import re
import sys
import textwrap


def wrap_md(text, width=120):
    out, para = [], []

    def flush():
        nonlocal para
        if not para:
            return
        first = para[0]
        m = re.match(r'^(\s*(?:[-*]|\d+\.)\s+)', first)
        lead = re.match(r'^(\s*)', first).group(1)
        indent = ' ' * len(m.group(1)) if m else lead
        joined = ' '.join(l.strip() for l in para)
        out.extend(textwrap.wrap(joined, width=width, initial_indent=lead, subsequent_indent=indent, break_long_words=False, break_on_hyphens=False))
        para = []
    incode = False
    for line in text.split('\n'):
        if line.strip().startswith('```'):
            flush(); incode = not incode; out.append(line); continue
        if incode or line.startswith('|') or line.startswith('#') or not line.strip():
            flush(); out.append(line); continue
        if re.match(r'^\s*(?:[-*]|\d+\.)\s+', line):
            flush()
        para.append(line)
    flush()
    return '\n'.join(out)


if __name__ == "__main__":
    for p in sys.argv[1:]:
        s = open(p).read()
        open(p, 'w').write(wrap_md(s))
