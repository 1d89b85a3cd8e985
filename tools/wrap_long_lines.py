#!/usr/bin/env python
"""Wrap source lines longer than LIMIT without changing the program: a trailing comment moves above its line; otherwise the line breaks after the last comma (inside
brackets) or inside a plain string literal (adjacent-literal concatenation, inside brackets) that fits. The module's AST is compared before and after - any
difference aborts the rewrite of that file. (No formatter is installed in this image.)"""
import ast
import io
import sys
import tokenize

LIMIT = 160


def split_line(lines, i):
    line = lines[i]
    indent = len(line) - len(line.lstrip())
    src = "".join(lines)
    toks = list(tokenize.generate_tokens(io.StringIO(src).readline))
    row = i + 1
    on = [t for t in toks if t.start[0] == row or (t.start[0] < row <= t.end[0])]
    # 1. trailing comment -> above
    for t in on:
        if t.type == tokenize.COMMENT and t.start[0] == row and t.start[1] > indent:
            code = line[:t.start[1]].rstrip()
            if code:
                lines[i] = code + "\n"
                lines.insert(i, " " * indent + t.string + "\n")
                return True
    # bracket depth at each token start on this row
    depth, best = 0, None
    for t in toks:
        if t.start[0] > row:
            break
        if t.type == tokenize.OP and t.string in "([{":
            depth += 1
        elif t.type == tokenize.OP and t.string in ")]}":
            depth -= 1
        if t.start[0] != row:
            continue
        if t.type == tokenize.OP and t.string == "," and depth > 0 and indent + 8 < t.end[1] <= LIMIT - 2:
            best = ("comma", t.end[1])
        if t.type == tokenize.STRING and depth > 0 and t.end[0] == row and t.string[0] in "\"'" and not t.string.startswith(('"""', "'''")):
            q = t.string[0]
            body_start, body_end = t.start[1] + 1, t.end[1] - 1
            cut = line.rfind(" ", body_start + 20, min(body_end - 10, LIMIT - 4))
            if cut > 0 and "\\" not in line[cut - 2:cut + 2] and (best is None or cut > best[1]):
                best = ("string", cut + 1, q)
    if best is None:
        return False
    cont = " " * (indent + 8)
    if best[0] == "comma":
        c = best[1]
        lines[i] = line[:c].rstrip() + "\n"
        lines.insert(i + 1, cont + line[c:].lstrip())
    else:
        c, q = best[1], best[2]
        lines[i] = line[:c] + q + "\n"
        lines.insert(i + 1, cont + q + line[c:])
    return True


def wrap(path):
    text = open(path).read()
    before = ast.dump(ast.parse(text))
    lines = text.splitlines(keepends=True)
    for _ in range(400):
        long = [k for k, l in enumerate(lines) if len(l.rstrip("\n")) > LIMIT]
        if not long:
            break
        progressed = False
        for k in long:
            try:
                if split_line(lines, k):
                    progressed = True
                    break
            except tokenize.TokenError:
                continue
        if not progressed:
            break
    out = "".join(lines)
    if ast.dump(ast.parse(out)) != before:
        print("%s: AST changed - not written" % path)
        return
    if out != text:
        open(path, "w").write(out)
    left = sum(1 for l in out.splitlines() if len(l) > LIMIT)
    print("%s: %d lines over %d left" % (path, left, LIMIT))


if __name__ == "__main__":
    for p in sys.argv[1:]:
        wrap(p)
