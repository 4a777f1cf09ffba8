#!/usr/bin/env python3
"""Writes the engine's knob registry (csrc/knobs.cpp, through mcgpu_knob_table) as the Markdown table of INTEGRATION.md section 6,
between the markers `<!-- knobs:begin -->` / `<!-- knobs:end -->`.  `--check` only compares (tests/test_formats_and_abi.py)."""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
TYPES = {"i": "integer", "f": "seconds", "b": "switch (set = on)", "s": "string"}
SCOPES = {"K": "kernel variant / schedule", "H": "host pipeline", "T": "test hook", "P": "Python side"}

def table(engine) -> str:
    rows = ["| variable | type | scope | default | what it selects |", "|---|---|---|---|---|"]
    for k in engine.knob_table():
        rows.append(f"| `{k['name']}` | {TYPES[k['type']]} | {SCOPES[k['scope']]} | {k['default']} | {k['what']} |")
    return "\n".join(rows) + "\n"

def main():
    import cases
    text = (ROOT / "INTEGRATION.md").read_text()
    a, b = text.index("<!-- knobs:begin -->\n") + len("<!-- knobs:begin -->\n"), text.index("<!-- knobs:end -->")
    new = text[:a] + table(cases.pkg.engine) + text[b:]
    if "--check" in sys.argv:
        sys.exit(0 if new == text else 1)
    (ROOT / "INTEGRATION.md").write_text(new)

if __name__ == "__main__":
    main()
