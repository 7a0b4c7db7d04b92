#!/usr/bin/env python3
"""Wraps the lines of C++ / HIP sources at a column limit without changing a token (the image has no clang-format): comment lines are re-flowed
at word boundaries, a trailing `// comment` that pushes a line over the limit moves to its own line(s) above it, and code is broken after the
last `, ` / ` && ` / ` || ` / ` ? ` / ` : ` / ` = ` / `; ` / `) ` / `{ ` that lies outside string and character literals and before the limit.
Preprocessor lines and lines that cannot be broken are left alone (reported).  usage: wrap_lines.py [--limit 160] files..."""
import re
import sys


def split_code_comment(line):
    """(code, comment or None): the `//` that starts a trailing comment, outside literals"""
    i, n, quote = 0, len(line), None
    while i < n:
        ch = line[i]
        if quote:
            if ch == "\\":
                i += 2
                continue
            if ch == quote:
                quote = None
        elif ch in "\"'":
            quote = ch
        elif ch == "/" and i + 1 < n and line[i + 1] == "/":
            return line[:i], line[i:]
        elif ch == "/" and i + 1 < n and line[i + 1] == "*":
            j = line.find("*/", i + 2)
            i = (j + 2) if j >= 0 else n
            continue
        i += 1
    return line, None


def wrap_comment(indent, text, limit):
    """text: the comment without its leading `//` marker; keeps leading spaces of the text (lists, continuation indents)"""
    m = re.match(r"^(\s*)(.*)$", text)
    lead, body = m.group(1), m.group(2)
    words = body.split(" ")
    out, cur = [], indent + "//" + lead
    first = True
    for w in words:
        if len(cur) + (0 if first else 1) + len(w) > limit and not first:
            out.append(cur.rstrip())
            cur = indent + "//" + lead + w
        else:
            cur += ("" if first else " ") + w
        first = False
    out.append(cur.rstrip())
    return out


def break_points(code):
    """indices AFTER which the code may be broken (outside literals and block comments), with a preference rank (lower = better)"""
    pts, i, n, quote, depth = [], 0, len(code), None, 0
    while i < n:
        ch = code[i]
        if quote:
            if ch == "\\":
                i += 2
                continue
            if ch == quote:
                quote = None
            i += 1
            continue
        if ch in "\"'":
            quote = ch
        elif ch == "/" and i + 1 < n and code[i + 1] == "*":
            j = code.find("*/", i + 2)
            i = (j + 2) if j >= 0 else n
            continue
        elif ch in "([{":
            depth += 1
        elif ch in ")]}":
            depth -= 1
        if ch == ";" and i + 1 < n and code[i + 1] == " ":
            pts.append((i + 1, 0, depth))
        elif ch == "{" and i + 1 < n and code[i + 1] == " ":
            pts.append((i + 1, 1, depth))
        elif ch == "," and i + 1 < n and code[i + 1] == " ":
            pts.append((i + 1, 2, depth))
        elif code.startswith(" && ", i) or code.startswith(" || ", i):
            pts.append((i + 3, 2, depth))
        elif code.startswith(" ? ", i) or code.startswith(" : ", i):
            pts.append((i + 2, 3, depth))
        elif code.startswith(" = ", i) or code.startswith(" + ", i):
            pts.append((i + 2, 4, depth))
        elif ch == ")" and i + 1 < n and code[i + 1] == " " and not code.startswith(") {", i):
            pts.append((i + 1, 5, depth))
        i += 1
    return pts


def wrap_code(code, limit, cont_indent):
    out = []
    while len(code) > limit:
        pts = [p for p in break_points(code) if len(cont_indent) + 8 < p[0] <= limit - 1 and code[p[0]:].strip()]
        if not pts:
            return None
        # the rightmost point among the best-ranked that still fills the line reasonably (> 60 % of the limit), else the rightmost of all
        good = [p for p in pts if p[0] > 0.6 * limit]
        cand = good or pts
        best_rank = min(p[1] for p in cand)
        pos = max(p[0] for p in cand if p[1] <= max(best_rank, 2))
        out.append(code[:pos].rstrip())
        code = cont_indent + code[pos:].lstrip()
    out.append(code)
    return out


def wrap_file(path, limit):
    lines = open(path).read().split("\n")
    out, failed = [], 0
    in_macro = False
    for line in lines:
        cont = in_macro
        in_macro = line.rstrip().endswith("\\")
        if len(line) <= limit or cont or in_macro or line.lstrip().startswith("#"):
            out.append(line)
            continue
        indent = re.match(r"^\s*", line).group(0)
        stripped = line.lstrip()
        if stripped.startswith("//"):
            out += wrap_comment(indent, stripped[2:], limit)
            continue
        code, comment = split_code_comment(line)
        if comment is not None:
            out += wrap_comment(indent, comment[2:], limit)       # the trailing comment goes above its line
            code = code.rstrip()
            if len(code) <= limit:
                out.append(code)
                continue
        wrapped = wrap_code(code, limit, indent + "        ")
        if wrapped is None:
            failed += 1
            out.append(code)
        else:
            out += wrapped
    open(path, "w").write("\n".join(out))
    return failed


def main():
    args = sys.argv[1:]
    limit = 160
    if args and args[0] == "--limit":
        limit = int(args[1]); args = args[2:]
    for path in args:
        failed = wrap_file(path, limit)
        longest = max(len(l) for l in open(path).read().split("\n"))
        print(f"{path}: longest line now {longest}" + (f", {failed} line(s) could not be broken" if failed else ""))


if __name__ == "__main__":
    main()
