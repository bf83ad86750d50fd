#!/usr/bin/env python3
"""Strict applier for a single-file unified diff (no fuzz, no offsets): fallback of robustvlm_amd/csrc/Makefile where patch(1) is
missing.  Usage: apply_unified_patch.py ORIGINAL PATCH OUTPUT"""
import re
import sys


def apply(orig_lines, patch_lines):
    out, pos, i = [], 0, 0
    while i < len(patch_lines) and not patch_lines[i].startswith("@@"):
        i += 1
    while i < len(patch_lines):
        m = re.match(r"@@ -(\d+)(?:,(\d+))? \+(\d+)(?:,(\d+))? @@", patch_lines[i])
        if not m:
            raise SystemExit(f"bad hunk header at patch line {i + 1}: {patch_lines[i]!r}")
        start = int(m.group(1)) - 1 if (m.group(2) is None or int(m.group(2)) > 0) else int(m.group(1))
        if start < pos:
            raise SystemExit(f"overlapping hunk at patch line {i + 1}")
        out.extend(orig_lines[pos:start])
        pos = start
        i += 1
        while i < len(patch_lines) and not patch_lines[i].startswith("@@"):
            line = patch_lines[i]
            tag, body = line[:1], line[1:]
            if tag in (" ", "-"):
                if pos >= len(orig_lines) or orig_lines[pos] != body:
                    raise SystemExit(f"patch does not apply at original line {pos + 1} (patch line {i + 1})")
                if tag == " ":
                    out.append(body)
                pos += 1
            elif tag == "+":
                out.append(body)
            elif tag == "\\":
                pass
            else:
                raise SystemExit(f"unexpected patch line {i + 1}: {line!r}")
            i += 1
    out.extend(orig_lines[pos:])
    return out


if __name__ == "__main__":
    o, p, dst = sys.argv[1:4]
    res = apply(open(o).read().split("\n"), open(p).read().split("\n")[:-1] if open(p).read().endswith("\n") else open(p).read().split("\n"))
    open(dst, "w").write("\n".join(res))
