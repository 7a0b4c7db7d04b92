#!/usr/bin/env python3
"""Re-flows the paragraphs of a Markdown file at a column limit (default 160): prose and list items are wrapped at word boundaries with hanging
indents kept; headings, table rows, fenced code blocks and HTML lines are left alone (and reported when they are over the limit).
usage: wrap_md.py [--limit 160] files..."""
import re
import sys
import textwrap


def reflow(text, limit):
    out, para, in_code = [], [], False

    def flush():
        if not para:
            return
        first = para[0]
        m = re.match(r"^(\s*(?:[-*+]|\d+\.)\s+)", first)
        lead = m.group(1) if m else re.match(r"^(\s*)", first).group(1)
        hang = " " * len(lead)
        body = " ".join([first[len(lead):].strip()] + [ln.strip() for ln in para[1:]])
        out.extend(textwrap.wrap(body, width=limit, initial_indent=lead, subsequent_indent=hang, break_long_words=False, break_on_hyphens=False) or [lead.rstrip()])
        para.clear()

    for ln in text.split("\n"):
        if ln.lstrip().startswith("```"):
            flush(); in_code = not in_code; out.append(ln); continue
        if in_code or ln.lstrip().startswith(("|", "#", "<")) or not ln.strip():
            flush(); out.append(ln); continue
        if re.match(r"^\s*(?:[-*+]|\d+\.)\s+", ln):      # a new list item starts a new paragraph
            flush()
        para.append(ln)
    flush()
    return "\n".join(out)


def main():
    args = sys.argv[1:]
    limit = 160
    if args[:1] == ["--limit"]:
        limit = int(args[1]); args = args[2:]
    for path in args:
        src = open(path).read()
        new = reflow(src, limit)
        if new != src:
            open(path, "w").write(new)
        over = [i + 1 for i, ln in enumerate(new.split("\n")) if len(ln) > limit]
        print(f"{path}: {len(new)} bytes, longest line {max(len(ln) for ln in new.split(chr(10)))}" + (f", over the limit at lines {over[:8]}" if over else ""))


if __name__ == "__main__":
    main()
