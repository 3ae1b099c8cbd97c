"""Re-wraps the prose of a Markdown file at a column limit (default 140): paragraphs and list items (continuation lines aligned with the item's
text); headings, tables, code fences and reference-style lines are left alone.  Prints what is still longer than the limit.
    python tools/wrap_md.py DESIGN.md [140]"""
import re
import sys
import textwrap

path = sys.argv[1]
limit = int(sys.argv[2]) if len(sys.argv) > 2 else 140
lines = open(path, encoding="utf-8").read().split("\n")
out, i, fence = [], 0, False
item = re.compile(r"^(\s*)([*+-]|\d+\.)\s+")


def flush(block):
    if not block:
        return
    first = block[0]
    m = item.match(first)
    if m:
        lead = m.group(0)
        indent = " " * len(lead)
    else:
        lead = re.match(r"^\s*", first).group(0)
        indent = lead
    text = " ".join(l.strip() for l in block)
    text = text[len(lead.strip()) + 1:] if m else text
    # a continuation line must not START like a list item ("- ", "+ ", "1. "): the space in front of such a token does not break
    text = re.sub(r" (?=(?:[*+-]|\d+\.) )", "\u00a0", text.strip())
    body = textwrap.wrap(text, width=limit, initial_indent=lead, subsequent_indent=indent, break_long_words=False, break_on_hyphens=False)
    body = [b.replace("\u00a0", " ") for b in body]
    out.extend(body)


block = []
while i < len(lines):
    l = lines[i]
    if l.strip().startswith("```"):
        flush(block); block = []
        fence = not fence
        out.append(l)
    elif fence or l.startswith("|") or l.startswith("#") or not l.strip():
        flush(block); block = []
        out.append(l)
    elif item.match(l):
        flush(block); block = [l]
    elif block and (len(l) - len(l.lstrip())) < (len(re.match(r"^\s*", block[0]).group(0))):      # dedent: a new paragraph after a nested item
        flush(block); block = [l]
    else:
        block.append(l)
    i += 1
flush(block)
open(path, "w", encoding="utf-8").write("\n".join(out))
long = [(n + 1, len(l)) for n, l in enumerate(out) if len(l) > limit]
print(f"{path}: {len(out)} lines, {len(long)} longer than {limit}" + (": " + ", ".join(f"{n} ({c})" for n, c in long[:40]) if long else ""))
