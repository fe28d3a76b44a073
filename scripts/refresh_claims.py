"""After profiles/ has been refreshed (scripts/r04_collect.py): carries the checked claims  ⟨value · profiles/<bench json>⟩  of the
documents over to the new session.  For every such claim the value is looked up in the OLD file (git HEAD) to find the JSON
path(s) it came from; the same path of the NEW file (working tree) gives the new value.  Claims whose value is found under
several paths with different new values, or under none, are reported and left alone (as are numbers in running prose that
were derived by hand: tests/test_docs.py does not check those, a reader has to).

    python scripts/refresh_claims.py [--apply]
"""
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DOCS = ["DESIGN.md", "README.md", "INTEGRATION.md", os.path.join("scripts", "README.md")]
CLAIM = re.compile(r"⟨([^⟩·]+)·\s*(profiles/[A-Za-z0-9_./-]+\.json)\s*⟩")
NUMBER = re.compile(r"[-+]?\d+(?:\.\d+)?")


def flatten(obj, path="", out=None):
    out = {} if out is None else out
    if isinstance(obj, dict):
        for k, v in obj.items():
            flatten(v, f"{path}/{k}", out)
    elif isinstance(obj, list):
        for i, v in enumerate(obj):
            flatten(v, f"{path}/{i}", out)
    elif isinstance(obj, (int, float)) and not isinstance(obj, bool):
        out[path] = obj
    return out


def load_line(text):
    lines = [l for l in text.strip().splitlines() if l.startswith("{") or l.startswith("[")]
    return json.loads(lines[-1]) if len(lines) == 1 or lines[-1].startswith("{") and not text.strip().startswith("[") else json.loads(text)


def main():
    apply = "--apply" in sys.argv
    cache = {}
    problems = changed = 0
    for doc in DOCS:
        path = os.path.join(ROOT, doc)
        text = open(path, encoding="utf-8").read()

        def repl(m):
            nonlocal problems, changed
            value, prof = m.group(1), m.group(2)
            if prof not in cache:
                old = subprocess.run(["git", "show", f"HEAD:{prof}"], capture_output=True, text=True, cwd=ROOT).stdout
                new = open(os.path.join(ROOT, prof), encoding="utf-8").read()
                try:
                    cache[prof] = (flatten(load_line(old)), flatten(load_line(new)))
                except Exception as e:                       # not a JSON line: nothing to carry over
                    cache[prof] = None
            if cache[prof] is None:
                return m.group(0)
            old, new = cache[prof]
            nums = NUMBER.findall(value)
            if len(nums) != 1:
                return m.group(0)
            tok = nums[0]
            dec = len(tok.split(".")[1]) if "." in tok else 0
            q = float(tok)
            paths = [p for p, v in old.items() if abs(round(v, dec) - q) <= 0.5 * 10 ** (-dec) * 1e-6 + 1e-12]
            news = {round(new[p], dec) for p in paths if p in new}
            if len(news) != 1:
                if abs(q) > 0 and not any(abs(round(v, dec) - q) <= 1e-12 for v in new.values()):
                    print(f"{doc}: ⟨{value.strip()} · {prof}⟩: {'no' if not paths else len(paths)} source path(s) {paths[:3]} -> {sorted(news)[:4]}: left alone")
                    problems += 1
                return m.group(0)
            nv = news.pop()
            ntok = f"{nv:.{dec}f}" if dec else str(int(nv))
            if ntok == tok:
                return m.group(0)
            changed += 1
            return m.group(0).replace(tok, ntok, 1)

        new_text = CLAIM.sub(repl, text)
        if apply and new_text != text:
            open(path, "w", encoding="utf-8").write(new_text)
    print(f"{changed} claim(s) {'rewritten' if apply else 'would change'}, {problems} left alone")


if __name__ == "__main__":
    main()
