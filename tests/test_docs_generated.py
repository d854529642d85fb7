"""CPU: the parity table and the figure table of DESIGN.md / README.md are rendered from the tracked evidence files of the latest round
(profiles/rNN_parity_*.json, profiles/rNN_bench_line_final.json) by tools/render_design_tables.py; this test re-renders them and fails
when a document differs -- a number in those tables cannot drift from the measurement."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def test_generated_tables_match_the_evidence_files():
    import render_design_tables as R
    ev, src = R.load()
    seen = 0
    for doc in R.DOCS:
        text = open(os.path.join(ROOT, doc)).read()
        seen += text.count("<!-- generated:")
        assert R.render(text, ev, src) == text, f"{doc}: generated tables are stale -- run tools/render_design_tables.py"
    assert seen >= 6, "DESIGN.md / README.md must carry the generated parity and figure tables (begin + end markers)"


def test_the_renderer_reads_one_round():
    """all three evidence files come from the same (latest) round: a table must not mix rounds"""
    import re
    import render_design_tables as R
    _ev, src = R.load()
    rounds = {re.match(r"profiles/r(\d+)_", p).group(1) for p in src.values()}
    assert len(rounds) == 1, src
