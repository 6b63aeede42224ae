#!/usr/bin/env python3
"""Every number a "Measured" table of DESIGN.md quotes must be greppable in the profiles/ file(s) the same row cites.

Convention (sections whose heading starts with "Measured"): a Markdown table whose LAST column is the evidence -- one or
more file names under profiles/ in backticks -- and whose other cells hold the quoted numbers.  For every such row this script takes
each number with a decimal point (31.706, 0.711, 1.59 ...; integers, shapes like 64x8x1024x1024 and numbers inside backticks or
parentheses marked "(derived)" are not checked) and looks for it, as text, in the cited files.  JSON evidence is searched both raw and
with every float re-printed at the precision the table uses (a bench line stores 31.405981 where the table says 31.41).

usage: tools/check_design_numbers.py [DESIGN.md]      exit code 1 and one line per miss if a number cannot be found."""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NUM = re.compile(r"(?<![\w.])(\d+\.\d+)(?![\w.]*\.\w)")


def floats_of(obj, out):
    if isinstance(obj, float):
        out.append(obj)
    elif isinstance(obj, dict):
        for v in obj.values():
            floats_of(v, out)
    elif isinstance(obj, (list, tuple)):
        for v in obj:
            floats_of(v, out)


def file_has(path, token):
    try:
        txt = open(path, errors="replace").read()
    except OSError:
        return None
    if re.search(r"(?<![\d.])" + re.escape(token) + r"(?!\d)", txt):
        return True
    nd = len(token.split(".")[1])
    vals = []
    if path.endswith(".json"):
        for line in txt.splitlines():
            line = line.strip()
            if line.startswith("{") or line.startswith("["):
                try:
                    floats_of(json.loads(line), vals)
                except ValueError:
                    pass
        if not vals:
            try:
                floats_of(json.loads(txt), vals)
            except ValueError:
                pass
    else:       # text / csv evidence: any longer decimal that rounds to the quoted one
        vals = [float(m) for m in re.findall(r"(?<![\w.])\d+\.\d+(?:[eE][-+]?\d+)?", txt)]
        if path.endswith(".csv"):
            # rocprofv3 kernel_stats.csv stores nanoseconds; a table may quote them as microseconds / milliseconds.  Only the DURATION columns
            # (AverageNs, MinNs, MaxNs) are rescaled -- round-5 advice: rescaling every integer of four or more digits (calls, totals, percentages)
            # made almost any quoted number "match"
            import csv as _csv
            dur = []
            try:
                rows = list(_csv.reader(txt.splitlines()))
                cols = [i for i, h in enumerate(rows[0]) if any(k in h for k in ("AverageNs", "MinNs", "MaxNs"))] if rows else []
                for r in rows[1:]:
                    for i in cols:
                        if i < len(r):
                            try:
                                dur.append(float(r[i]))
                            except ValueError:
                                pass
            except Exception:
                dur = []
            vals += [v * 1e-3 for v in dur] + [v * 1e-6 for v in dur]
    want = float(token)
    return any(abs(round(v, nd) - want) < 0.5 * 10 ** (-nd) * 1e-6 + 1e-12 or ("%.*f" % (nd, v)) == token for v in vals)


def main():
    path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "DESIGN.md")
    lines = open(path).read().splitlines()
    in_measured, misses, checked, rows = False, [], 0, 0
    for ln, line in enumerate(lines, 1):
        if line.startswith("#"):
            in_measured = re.sub(r"^[\d.\s]+", "", line.lstrip("#").strip()).lower().startswith("measured")
            continue
        if not in_measured or not line.startswith("|") or re.match(r"^\|\s*-", line):
            continue
        cells = [c.strip() for c in line.strip().strip("|").split("|")]
        if len(cells) < 2:
            continue
        files = []
        for f in re.findall(r"`([^`]+\.(?:txt|json|csv))`", cells[-1]):      # `a_{x,y}.json` stands for a_x.json and a_y.json
            m = re.search(r"\{([^{}]+)\}", f)
            files += [f[:m.start()] + alt + f[m.end():] for alt in m.group(1).split(",")] if m else [f]
        if not files:
            continue                      # header row or a row without evidence files
        rows += 1
        body = " | ".join(cells[:-1])
        body = re.sub(r"`[^`]*`", " ", body)                       # code spans: names, flags, shapes
        body = re.sub(r"\([^()]*derived[^()]*\)", " ", body)       # "(derived: ...)" arithmetic on checked numbers
        paths = [os.path.join(ROOT, "profiles", os.path.basename(f)) for f in files]
        for tok in NUM.findall(body):
            checked += 1
            res = [file_has(p, tok) for p in paths]
            if any(r is None for r in res):
                misses.append("%s:%d: evidence file missing: %s" % (os.path.basename(path), ln, [f for f, r in zip(files, res) if r is None]))
            if not any(res):
                misses.append("%s:%d: %s not found in %s" % (os.path.basename(path), ln, tok, ", ".join(files)))
    for m in misses:
        print(m)
    print("%d numbers in %d evidence rows checked, %d misses" % (checked, rows, len(misses)))
    return 1 if misses else 0


if __name__ == "__main__":
    sys.exit(main())
