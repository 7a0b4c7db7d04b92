"""tools/wrap_lines.py re-flowed the host translation units of round 5 at 160 columns (the image has no clang-format).  It must never change a token:
checked here on a synthetic source with the constructs those files contain (string literals with `//` and `, ` inside, trailing comments, block comments,
preprocessor lines, backslash-continued macros, long conditions), and on the product's own host sources, which must already be at rest under it."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOOL = os.path.join(ROOT, "tools", "wrap_lines.py")
HOST_SOURCES = ["engine.h", "workspace.hip", "verify_stages.hip", "msm_ops.hip", "host_pipeline.hip", "options.hip", "multi_device.hip", "entry_points.hip"]


def code_tokens(text):
    """the text with comments removed and all whitespace dropped outside string / character literals"""
    out, i, n, quote = [], 0, len(text), None
    while i < n:
        ch = text[i]
        if quote:
            out.append(ch)
            if ch == "\\":
                out.append(text[i + 1]); i += 2
                continue
            if ch == quote:
                quote = None
        elif ch in "\"'":
            quote = ch; out.append(ch)
        elif text.startswith("//", i):
            j = text.find("\n", i)
            i = n if j < 0 else j
            continue
        elif text.startswith("/*", i):
            j = text.find("*/", i + 2)
            i = n if j < 0 else j + 2
            continue
        elif not ch.isspace():
            out.append(ch)
        i += 1
    return "".join(out)


def comment_words(text):
    return re.findall(r"[A-Za-z0-9_]+", " ".join(re.findall(r"//(.*)", text)))


def test_wrapping_preserves_every_token_and_every_comment_word(tmp_path):
    long_args = ", ".join(f"argument_number_{i}" for i in range(30))
    src = "\n".join([
        "#include <cstdio>",
        "#define LONG_MACRO(x)                                                                                                                                                  \\",
        "    do { (void)(x); } while (0)",
        f"int f({', '.join('int a%d' % i for i in range(40))}) {{ return 0; }}      // a trailing comment that is long enough to push the line over the limit all by itself, yes",
        f'static const char *s = "a string, with // two slashes, and commas, {"x" * 120}";   // and a comment after it',
        f"    if ((rc = g(1)) || (rc = g(22222222)) || (rc = g(33333333333)) || (rc = g(4444444444444)) || (rc = g(5555555555555555)) || (rc = g(66666666666666666)) || (rc = g(7))) return rc;",
        f"    call_something({long_args});",
        "// " + " ".join(f"word{i}" for i in range(80)),
        "    x = cond ? " + " + ".join(f"term_{i}" for i in range(40)) + " : 0;   /* block */",
        "",
    ])
    p = tmp_path / "sample.hip"
    p.write_text(src)
    subprocess.run([sys.executable, TOOL, "--limit", "120", str(p)], check=True, capture_output=True)
    out = p.read_text()
    assert code_tokens(out) == code_tokens(src)
    assert sorted(comment_words(out)) == sorted(comment_words(src))
    long_lines = [ln for ln in out.split("\n") if len(ln) > 120 and not ln.lstrip().startswith("#") and not ln.rstrip().endswith("\\") and '"' not in ln]
    assert not long_lines, long_lines
    again = tmp_path / "again.hip"
    again.write_text(out)
    subprocess.run([sys.executable, TOOL, "--limit", "120", str(again)], check=True, capture_output=True)
    assert again.read_text() == out                               # at rest after one pass


def test_the_host_sources_are_at_rest_and_within_160_columns(tmp_path):
    for name in HOST_SOURCES:
        src = open(os.path.join(ROOT, "kzg_rust_amd", "csrc", name)).read()
        assert max(len(ln) for ln in src.split("\n")) <= 160, name
        p = tmp_path / name
        p.write_text(src)
        subprocess.run([sys.executable, TOOL, str(p)], check=True, capture_output=True)
        assert p.read_text() == src, f"{name} changes under tools/wrap_lines.py"


def test_design_md_is_within_its_limits():
    """DESIGN.md is the current design only (history lives in EXPERIMENTS.md): at most 40 KB and every line within 160 BYTES (prose re-flowed by tools/wrap_md.py)"""
    raw = open(os.path.join(ROOT, "DESIGN.md"), "rb").read()
    assert len(raw) <= 40 * 1024, len(raw)
    over = [i + 1 for i, ln in enumerate(raw.split(b"\n")) if len(ln) > 160]
    assert not over, over
    assert os.path.exists(os.path.join(ROOT, "EXPERIMENTS.md"))
